"""Drop-in for the reference's src/parallel_trainer.py: self-play driver, replay
buffer, training step and the two on-disk formats.

What differs from the reference in structure (results are the same):

  * examples are held as three contiguous float32 arrays (`ExampleArrays`, a
    sequence of (state, policy, wl) tuples to its users) instead of a Python
    list of small arrays; in `learn()` they are uploaded to the training device
    once and every batch is an index gather there, the mirror augmentation
    (src/parallel_trainer.py:33-70) being a column flip and a fixed permutation
    of the 192 policy entries applied to the rows that ask for it;
  * batch order comes from torch's own DataLoader over the index range
    [0, 2n), which draws from the global torch RNG exactly as the reference's
    DataLoader over GameDataset does (:231-232), so a seeded run visits the
    same examples in the same order;
  * `execute_parallel_episodes` runs on the device engine (libazamd): fused
    search + move choice + sample queue when the model is an AlphaZeroNet the
    HIP inference supports, otherwise the reference's host loop over the
    drop-in MCTS with any duck-typed model.
"""
import glob
import os
import re
from typing import Iterable, List, Optional, Tuple

import numpy as np
import torch
import torch.optim as optim
from torch.optim.lr_scheduler import CosineAnnealingLR
from torch.utils.data import DataLoader, Dataset

from .config import Config
from .game import BOARD_SIZE, NUM_ACTIONS, WHITE, BreakthroughGame
from .mcts import MCTS, Node
from .model import AlphaZeroNet


def _mirror_permutation() -> np.ndarray:
    """perm with mirrored_policy = policy[perm]: the source square is the column
    mirror of the target square, forward stays forward and the two diagonals
    trade places (src/parallel_trainer.py:56-67).  It is its own inverse."""
    perm = np.empty(NUM_ACTIONS, np.int64)
    swap = (0, 2, 1)
    for target in range(NUM_ACTIONS):
        sq, d = divmod(target, 3)
        row, col = divmod(sq, BOARD_SIZE)
        perm[target] = (row * BOARD_SIZE + (BOARD_SIZE - 1 - col)) * 3 + swap[d]
    return perm


MIRROR_PERM = _mirror_permutation()


def _concrete(device) -> torch.device:
    """torch.device with the ordinal filled in ("cuda" -> "cuda:<current>"), so that it
    compares equal to the device of tensors that live there."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def augment_example(state: np.ndarray, policy: np.ndarray, wl: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Left-right mirror of one example (src/parallel_trainer.py:33-70): planes
    flipped along the column axis, policy permuted, wl unchanged."""
    return np.ascontiguousarray(state[:, :, ::-1]), policy[MIRROR_PERM], wl


class ExampleArrays:
    """n training examples as arrays states[n,3,8,8], policies[n,192], wls[n,2]
    (the layout of training_data.npz) that reads like the reference's list of
    (state, policy, wl) tuples: len, indexing, slicing, iteration, +.

    The three arrays may be given as numpy arrays or as torch tensors on any device.
    Examples that were produced on a GPU stay there: `.states`, `.policies` and `.wls`
    (numpy) are made on first use, `tensors(device)` hands out torch tensors without a
    host round trip when they already live on that device."""

    __slots__ = ("_np", "_t")

    def __init__(self, states=None, policies=None, wls=None):
        if torch.is_tensor(states):
            t = tuple(x.detach().to(torch.float32).contiguous() for x in (states, policies, wls))
            if not (t[0].shape[0] == t[1].shape[0] == t[2].shape[0]):
                raise ValueError("states, policies and wls must have the same length")
            self._t = (t[0].reshape(-1, 3, BOARD_SIZE, BOARD_SIZE), t[1].reshape(-1, NUM_ACTIONS), t[2].reshape(-1, 2))
            self._np = None
            return
        st = np.zeros((0, 3, BOARD_SIZE, BOARD_SIZE), np.float32) if states is None else np.asarray(states, np.float32)
        po = np.zeros((0, NUM_ACTIONS), np.float32) if policies is None else np.asarray(policies, np.float32)
        wl = np.zeros((0, 2), np.float32) if wls is None else np.asarray(wls, np.float32)
        if not (len(st) == len(po) == len(wl)):
            raise ValueError("states, policies and wls must have the same length")
        self._np = (st, po, wl)
        self._t = None

    def _load(self):
        """Hook of views that fetch their contents on first use (ReplayView)."""

    def _host(self):
        self._load()
        if self._np is None:
            self._np = tuple(x.cpu().numpy() for x in self._t)
        return self._np

    @property
    def states(self):
        return self._host()[0]

    @property
    def policies(self):
        return self._host()[1]

    @property
    def wls(self):
        return self._host()[2]

    def tensors(self, device):
        """(states, policies, wls) as float32 torch tensors on `device`."""
        device = _concrete(device)
        self._load()
        if self._t is not None and self._t[0].device == device:
            return self._t
        if self._t is not None and self._np is None:
            return tuple(x.to(device) for x in self._t)
        return tuple(torch.from_numpy(np.ascontiguousarray(x)).to(device) for x in self._host())

    @classmethod
    def of(cls, examples) -> "ExampleArrays":
        if isinstance(examples, ExampleArrays):
            return examples
        examples = list(examples)
        if not examples:
            return cls()
        return cls(np.stack([e[0] for e in examples]), np.stack([e[1] for e in examples]),
                   np.stack([e[2] for e in examples]))

    def __len__(self):
        self._load()
        return len(self._np[0]) if self._np is not None else int(self._t[0].shape[0])

    def __getitem__(self, i):
        self._load()
        if isinstance(i, slice):
            if self._np is None:
                return ExampleArrays(self._t[0][i], self._t[1][i], self._t[2][i])
            return ExampleArrays(self.states[i], self.policies[i], self.wls[i])
        return self.states[i], self.policies[i], self.wls[i]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __add__(self, other):
        return ExampleArrays.concat([self, ExampleArrays.of(other)])

    def extend(self, other):
        merged = self + other
        self._np, self._t = merged._np, merged._t

    @staticmethod
    def concat(parts: Iterable["ExampleArrays"]) -> "ExampleArrays":
        parts = [p for p in parts if len(p)]
        if not parts:
            return ExampleArrays()
        if len(parts) == 1:
            return parts[0]
        for p in parts:
            p._load()
        devs = {p._t[0].device for p in parts if p._t is not None and p._np is None}
        if len(devs) == 1 and all(p._t is not None for p in parts):
            dev = devs.pop()
            ts = [p.tensors(dev) for p in parts]
            return ExampleArrays(*(torch.cat([t[i] for t in ts]) for i in range(3)))
        return ExampleArrays(np.concatenate([p.states for p in parts]), np.concatenate([p.policies for p in parts]),
                             np.concatenate([p.wls for p in parts]))


class ReplayView(ExampleArrays):
    """What `trainer.examples` returns: the replay buffer read like the reference's list
    (src/parallel_trainer.py:127, 437-442).  `len()` is answered from the ring; the rows are
    copied out of it (oldest first, on the training device) only when they are first touched.
    `extend` and `+=` append to the ring itself, as they would to the reference's list."""

    __slots__ = ("_trainer",)

    def __init__(self, trainer):
        self._np = self._t = None
        self._trainer = trainer

    def _load(self):
        if self._np is None and self._t is None:
            snap = self._trainer._replay.ordered()
            self._np, self._t = snap._np, snap._t

    def __len__(self):
        if self._np is None and self._t is None:
            return len(self._trainer._replay)
        return ExampleArrays.__len__(self)

    def extend(self, other):
        self._trainer.add_examples(other)
        self._np = self._t = None            # read again from the ring when next touched

    def __iadd__(self, other):
        self.extend(other)
        return self


class ReplayRing:
    """The replay buffer (src/parallel_trainer.py:437-442: append, keep the most recent
    `limit` examples) as a ring of rows in three tensors that live on the training device.
    New examples are written behind the newest row and overwrite the oldest once `limit`
    rows are held; nothing is moved or re-uploaded when the buffer is full.  Logical row i
    (0 = oldest) is physical row (start + i) mod rows."""

    def __init__(self, device):
        self.device = _concrete(device)
        self.s = self.p = self.w = None
        self.rows = 0          # allocated rows
        self.start = 0
        self.n = 0
        self.version = 0       # bumped by every change of the contents

    def __len__(self):
        return self.n

    def clear(self):
        self.s = self.p = self.w = None
        self.rows = self.start = self.n = 0
        self.version += 1

    def physical(self, logical):
        """Storage rows of logical rows (a tensor of indices on the ring's device)."""
        return (logical + self.start) % self.rows if self.start else logical

    def ordered(self) -> ExampleArrays:
        """A copy of the contents, oldest first (on the ring's device; later appends do not
        reach into it)."""
        if self.n == 0:
            return ExampleArrays()
        if self.start == 0:
            return ExampleArrays(self.s[:self.n].clone(), self.p[:self.n].clone(), self.w[:self.n].clone())
        idx = self.physical(torch.arange(self.n, device=self.device))
        return ExampleArrays(self.s[idx], self.p[idx], self.w[idx])

    def _rebuild(self, rows):
        """Storage of `rows` rows holding the newest min(n, rows) examples from row 0 on."""
        old = self.ordered() if self.n else None
        keep = min(self.n, rows)
        self.s = torch.empty((rows, 3, BOARD_SIZE, BOARD_SIZE), dtype=torch.float32, device=self.device)
        self.p = torch.empty((rows, NUM_ACTIONS), dtype=torch.float32, device=self.device)
        self.w = torch.empty((rows, 2), dtype=torch.float32, device=self.device)
        if keep:
            os_, op, ow = old.tensors(self.device)
            self.s[:keep], self.p[:keep], self.w[:keep] = os_[self.n - keep:], op[self.n - keep:], ow[self.n - keep:]
        self.rows, self.start, self.n = rows, 0, keep

    def append(self, examples: ExampleArrays, limit: int):
        limit = max(1, int(limit))
        k = len(examples)
        ts = examples.tensors(self.device) if k else None
        if k > limit:
            ts = tuple(t[k - limit:] for t in ts)
            k = limit
        if self.rows > limit:                        # the cap was lowered
            self._rebuild(limit)
        need = min(limit, self.n + k)
        if need > self.rows:                         # grow (doubling) until the cap is reached
            self._rebuild(min(limit, max(need, 2 * self.rows, 1024)))
        if self.n > limit:                           # cap lowered below the contents without new rows
            self.start = (self.start + self.n - limit) % self.rows
            self.n = limit
        if k:
            pos = (torch.arange(k, device=self.device) + (self.start + self.n)) % self.rows
            self.s.index_copy_(0, pos, ts[0])
            self.p.index_copy_(0, pos, ts[1])
            self.w.index_copy_(0, pos, ts[2])
            over = max(0, self.n + k - min(limit, self.rows))
            self.start = (self.start + over) % self.rows
            self.n = self.n + k - over
        self.version += 1


class GameDataset(Dataset):
    """Examples followed by their mirrored copies (src/parallel_trainer.py:73-98)."""

    def __init__(self, examples, augment: bool = True):
        self.examples = ExampleArrays.of(examples)
        self.augment = augment

    def __len__(self):
        return len(self.examples) * (2 if self.augment else 1)

    def __getitem__(self, idx):
        n = len(self.examples)
        if self.augment and idx >= n:
            item = augment_example(*self.examples[idx - n])
        else:
            item = self.examples[idx]
        return tuple(torch.tensor(np.ascontiguousarray(x), dtype=torch.float32) for x in item)


def gather_batch(states, policies, wls, idx, mirror_perm, n=None, physical=None):
    """Rows idx of the augmented data set [examples ; mirrored examples] from
    device-resident arrays; idx >= n addresses the mirrored copy of idx - n.
    `physical` maps example numbers to storage rows (a ReplayRing that has wrapped)."""
    n = states.shape[0] if n is None else n
    mirrored = idx >= n
    real = torch.where(mirrored, idx - n, idx)
    if physical is not None:
        real = physical(real)
    s, p, w = states[real], policies[real], wls[real]
    s = torch.where(mirrored[:, None, None, None], s.flip(3), s)
    p = torch.where(mirrored[:, None], p[:, mirror_perm], p)
    return s, p, w


class ParallelTrainer:
    """Same constructor, attributes and methods as the reference's trainer
    (src/parallel_trainer.py:101-476).  `seed` (extra) seeds the device RNG of
    the fused self-play path; `verbose` (extra) enables the progress prints;
    `autocast_dtype` (extra, default None = float32 like the reference) runs the
    training forward under torch.autocast."""

    def __init__(self, model: AlphaZeroNet, device: str = "cpu",
                 num_parallel_games: int = Config.PARALLEL_GAMES,
                 num_simulations: int = Config.MCTS_SIMULATIONS,
                 checkpoint_dir: str = Config.CHECKPOINT_DIR, seed: int = 0, verbose: bool = True,
                 autocast_dtype=None, graph_step: bool = False):
        self.model = model
        self.device = device
        self.num_parallel_games = num_parallel_games
        self.num_simulations = num_simulations
        self.optimizer = optim.Adam(model.parameters(), lr=Config.LEARNING_RATE, weight_decay=Config.WEIGHT_DECAY)
        self.scheduler = CosineAnnealingLR(self.optimizer, T_max=Config.LR_SCHEDULER_T_MAX,
                                           eta_min=Config.LR_SCHEDULER_ETA_MIN)
        self.mcts = MCTS(model, num_simulations=num_simulations, device=device)
        self._replay = ReplayRing(device)       # lives on the training device
        self._examples_view = (None, -1)        # (ExampleArrays, ring version) of the last `examples` read
        self.max_examples = Config.BUFFER_SIZE
        self.checkpoint_dir = checkpoint_dir
        self.seed = seed
        self.verbose = verbose
        self.autocast_dtype = autocast_dtype   # None: the reference's float32 step; torch.bfloat16: autocast forward
        # replay every full batch of learn() from captured HIP graphs (distributed.GraphedTrainStep):
        # worth it where a step is bound by its launches, i.e. at a data-parallel rank's slice
        self.graph_step = bool(graph_step)
        self._graphed = {}                     # slice size -> GraphedTrainStep
        self._selfplay = None          # fused device self-play, built on first use
        self._selfplay_key = None      # version counters of the weights its HIP copy was made from
        os.makedirs(self.checkpoint_dir, exist_ok=True)

    # ------------------------------------------------------------------ replay buffer
    @property
    def examples(self) -> ExampleArrays:
        """The replay buffer, oldest example first.  The rows live on the training device; `len()`
        costs nothing, the rows are copied out of the ring when first touched and the numpy arrays
        behind the view are made when they are first read."""
        view, version = self._examples_view
        if version != self._replay.version:
            view = ReplayView(self)
            self._examples_view = (view, self._replay.version)
        return view

    @examples.setter
    def examples(self, value):
        if isinstance(value, ReplayView) and value._trainer is self:
            return                           # `trainer.examples += x` hands the view back: already appended
        self._replay.clear()
        self._replay.append(ExampleArrays.of(value), self.max_examples)

    def add_examples(self, new_examples):
        """Append and keep the most recent `max_examples` (src/parallel_trainer.py:437-442).
        Examples that are already on the training device (what execute_parallel_episodes
        returns on the fused path) are copied there, not through the host."""
        self._replay.append(ExampleArrays.of(new_examples), self.max_examples)

    def _weights_key(self):
        m = self.model
        return tuple(t._version for t in list(m.parameters()) + list(m.buffers()))

    @property
    def _weights_changed(self) -> bool:
        """The fused self-play's bf16 copy of the weights is older than the module's tensors
        (every in-place update - an optimiser step, load_state_dict, a caller's own change -
        moves a tensor's version counter)."""
        return self._selfplay is not None and self._selfplay_key != self._weights_key()

    def _say(self, msg):
        if self.verbose:
            print(msg)

    # ------------------------------------------------------------------ self-play
    def _fused_ok(self) -> bool:
        m = self.model
        return (isinstance(m, AlphaZeroNet) and m.hip_supported() and str(self.device).startswith("cuda")
                and type(self.mcts) is MCTS and self.mcts.model is m)

    def execute_parallel_episodes(self, num_games: Optional[int] = None, continuous: bool = False) -> ExampleArrays:
        """num_games games from the start position to the end, tree reuse, root noise,
        temperature 1 for the first TEMPERATURE_THRESHOLD plies then 0
        (src/parallel_trainer.py:137-223).  Returns the (state, policy, wl) examples.
        continuous=True (extra; fused path only) keeps num_parallel_games games running all
        the time and returns once num_games of them have finished (SelfPlay.collect_games):
        games in progress carry over to the next call."""
        if num_games is None:
            num_games = self.num_parallel_games
        if self._fused_ok():
            from .selfplay import SelfPlay
            sp = self._selfplay
            slots = max(num_games, self.num_parallel_games) if continuous else num_games
            if sp is None or sp.engine.n < slots or sp.num_simulations != self.mcts.num_simulations:
                sp = self._selfplay = SelfPlay(self.model, n_games=slots,
                                               num_simulations=self.mcts.num_simulations,
                                               c_puct=self.mcts.c_puct, seed=self.seed,
                                               device=torch.device(self.device).index)
                self._selfplay_key = self._weights_key()
            if self._weights_changed:
                sp.inference.refresh()
                self._selfplay_key = self._weights_key()
            if continuous:
                return ExampleArrays(*sp.collect_games(num_games, device_output=True))
            return ExampleArrays(*sp.execute_parallel_episodes(num_games, device_output=True))
        if continuous:
            raise RuntimeError("continuous self-play runs on the fused engine only (a HIP-supported AlphaZeroNet on a GPU)")
        return self._host_loop_episodes(num_games)

    def _host_loop_episodes(self, num_games: int) -> ExampleArrays:
        """The reference's loop itself, over the drop-in MCTS (device trees, model called
        between tree kernels): exact replay of its numpy draws."""
        games = [BreakthroughGame() for _ in range(num_games)]
        roots: List[Optional[Node]] = [None] * num_games
        history = [[] for _ in range(num_games)]
        plies = [0] * num_games
        active = list(range(num_games))
        out_s, out_p, out_w = [], [], []
        while active:
            new_roots = self.mcts.search_batch([games[i] for i in active], [roots[i] for i in active], add_noise=True)
            still = []
            for i, root in zip(active, new_roots):
                game = games[i]
                temperature = 1.0 if plies[i] < Config.TEMPERATURE_THRESHOLD else 0.0
                probs = self.mcts.get_action_probs(root, temperature=temperature)
                history[i].append((game.get_encoded_state(), probs, game.turn))
                action = int(np.random.choice(len(probs), p=probs))
                game.step(game.decode_action(action))
                plies[i] += 1
                roots[i] = root.children[action] if action in root.children else None
                if game.is_terminal():
                    w, l = game.get_result()
                    for s, p, turn in history[i]:
                        out_s.append(s)
                        out_p.append(p)
                        out_w.append((w, l) if turn == WHITE else (l, w))
                else:
                    still.append(i)
            active = still
        if not out_s:
            return ExampleArrays()
        return ExampleArrays(np.stack(out_s), np.stack(out_p), np.array(out_w, np.float32))

    # ------------------------------------------------------------------ training
    def learn(self, epochs: int = 10, batch_size: int = 64):
        """epochs passes over [examples ; mirrored examples] in shuffled batches:
        soft-target policy cross-entropy + WL cross-entropy, gradient clipping, Adam;
        one cosine-schedule step at the end (src/parallel_trainer.py:225-284).
        With torch.distributed initialised every rank must hold the same buffer and
        RNG state; each then trains on its slice of every batch and gradients are
        averaged.  Returns the mean losses of the last epoch."""
        from . import distributed as azdist
        import torch.distributed as dist
        ring = self._replay
        n = len(ring)
        if n == 0:
            return None
        dev = ring.device
        states, policies, wls = ring.s, ring.p, ring.w     # gathered in place: no copy of the buffer
        perm = torch.from_numpy(MIRROR_PERM).to(dev)
        # same sampler, same draws from the global torch RNG as a DataLoader over the 2n examples
        index_batches = DataLoader(torch.arange(2 * n), batch_size=batch_size, shuffle=True)
        world, rank = 1, 0
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(), dist.get_rank()
        self._say(f"Training on {2 * n} examples (augmented), LR: {self.optimizer.param_groups[0]['lr']:.6f}")
        stats = None
        use_graph = self.graph_step and dev.type == "cuda"
        for _ in range(epochs):
            tot = torch.zeros(3, dtype=torch.float32, device=dev)     # summed on the device: one host read per epoch
            for g in self._graphed.values():
                g.sums.zero_()
            count = 0
            for idx in index_batches:
                global_batch = len(idx)
                if world > 1:
                    idx = idx[rank::world]
                s, p, w = gather_batch(states, policies, wls, idx.to(dev), perm, n=n, physical=ring.physical)
                if use_graph and global_batch == batch_size and len(idx) > 0:
                    step = self._graphed.get(len(idx))
                    if step is None:
                        step = self._graphed[len(idx)] = azdist.GraphedTrainStep(
                            self.model, self.optimizer, len(idx), autocast_dtype=self.autocast_dtype)
                    step(s, p, w, global_batch=global_batch)
                else:
                    tot += azdist.train_step(self.model, self.optimizer, s, p, w, autocast_dtype=self.autocast_dtype,
                                             global_batch=global_batch, as_tensor=True).to(dev)
                count += 1
            for g in self._graphed.values():
                tot += g.sums
            stats = dict(zip(("loss", "pi", "wl"), (tot.double().cpu().numpy() / max(count, 1)).tolist()))
        self.scheduler.step()
        return stats

    # ------------------------------------------------------------------ checkpoints
    def _optimizer_state(self):
        """optimizer.state_dict() as the reference's trainer writes and reads it: an optimiser
        made capturable for the graphed step keeps its step counters on the GPU and says so in
        its parameter groups, which an Adam on another kind of device refuses to load."""
        sd = self.optimizer.state_dict()
        if not any(g.get("capturable") for g in sd["param_groups"]):
            return sd
        groups = [dict(g, capturable=False) for g in sd["param_groups"]]
        state = {k: {n: (v.detach().cpu() if n == "step" and torch.is_tensor(v) else v) for n, v in st.items()}
                 for k, st in sd["state"].items()}
        return {"state": state, "param_groups": groups}

    def _checkpoint_dict(self):
        return {
            "state_dict": self.model.state_dict(),
            "optimizer": self._optimizer_state(),
            "scheduler": self.scheduler.state_dict(),
            "config": {"num_blocks": self.model.num_blocks, "num_filters": self.model.num_filters},
        }

    def _restore(self, path):
        # tensors, numbers, strings, lists and dicts only: the restricted loader reads it
        ck = torch.load(path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(ck["state_dict"])
        self.optimizer.load_state_dict(ck["optimizer"])
        self._graphed = {}                     # captured graphs hold the optimiser state that was just replaced
        if "scheduler" in ck:
            self.scheduler.load_state_dict(ck["scheduler"])
            self.scheduler.T_max = Config.LR_SCHEDULER_T_MAX   # the code's horizon wins over the file's

    def save_iteration_checkpoint(self, iteration: int):
        """iteration_N.pt with keys state_dict / optimizer / scheduler / iteration / config
        (src/parallel_trainer.py:290-311)."""
        ck = self._checkpoint_dict()
        ck["iteration"] = iteration
        name = f"iteration_{iteration}.pt"
        torch.save(ck, os.path.join(self.checkpoint_dir, name))
        self._say(f"Checkpoint saved: {name}")

    def load_iteration_checkpoint(self, iteration: int) -> bool:
        name = f"iteration_{iteration}.pt"
        path = os.path.join(self.checkpoint_dir, name)
        if not os.path.exists(path):
            self._say(f"No checkpoint found: {name}")
            return False
        self._restore(path)
        self._say(f"Loaded checkpoint: {name}")
        return True

    def get_latest_iteration(self) -> int:
        """Highest N among iteration_N.pt in the checkpoint directory, 0 if none."""
        best = 0
        for f in glob.glob(os.path.join(self.checkpoint_dir, "iteration_*.pt")):
            m = re.search(r"iteration_(\d+)\.pt$", f)
            if m:
                best = max(best, int(m.group(1)))
        return best

    def save_checkpoint(self, filename: str = "checkpoint.pt"):
        """Legacy single-file form (src/parallel_trainer.py:448-460)."""
        path = os.path.join(self.checkpoint_dir, filename)
        torch.save(self._checkpoint_dict(), path)
        self._say(f"Checkpoint saved to {path}")

    def load_checkpoint(self, filename: str = "checkpoint.pt"):
        path = os.path.join(self.checkpoint_dir, filename)
        if os.path.exists(path):
            self._restore(path)
            self._say(f"Checkpoint loaded from {path}")
        else:
            self._say(f"No checkpoint found at {path}")

    # ------------------------------------------------------------------ training data file
    def _get_data_file_path(self) -> str:
        return os.path.join(self.checkpoint_dir, Config.DATA_FILE)

    def append_training_data(self, new_examples):
        """training_data.npz = arrays states / policies / wls of every example so far
        (src/parallel_trainer.py:367-396)."""
        new = ExampleArrays.of(new_examples)
        if len(new) == 0:
            return
        path = self._get_data_file_path()
        if os.path.exists(path):
            with np.load(path) as old:
                new = ExampleArrays(old["states"], old["policies"], old["wls"]) + new
        np.savez(path, states=new.states, policies=new.policies, wls=new.wls)
        self._say(f"Training data saved: {len(new)} total examples")

    def load_training_data(self, max_examples: Optional[int] = None) -> int:
        """The most recent max_examples of the data file become the replay buffer
        (src/parallel_trainer.py:398-435)."""
        if max_examples is None:
            max_examples = self.max_examples
        path = self._get_data_file_path()
        if not os.path.exists(path):
            self._say("No training data file found")
            self._replay.clear()
            return 0
        with np.load(path) as data:
            total = len(data["states"])
            start = max(0, total - max_examples)
            self._replay.clear()
            self._replay.append(ExampleArrays(data["states"][start:], data["policies"][start:], data["wls"][start:]),
                                max_examples)
        self._say(f"Loaded {len(self._replay)} training examples (of {total} total on disk)")
        return len(self._replay)
