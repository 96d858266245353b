"""Batched PUCT MCTS with the call surface of the reference's src/mcts.py, running on
the device search engine.

`MCTS(model, ...)`.search_batch(games, roots, add_noise) returns `Node` objects
as the reference does.  The trees live in HBM (one engine slot per game); a
`Node` is a handle that reads a node's statistics from the device on demand.
There is no CPU search: without a GPU or without libazamd every search raises.

Two evaluation paths:
  * `model` is an `AlphaZeroNet` the HIP kernels support, on a GPU device: the
    whole search (tree kernels + bf16 network) runs inside libazamd
    (az_engine_search).
  * any other object with `.eval()` and `__call__` (src/mcts.py:248-250), on any
    device: the tree kernels run on the GPU and `model` is called between
    them exactly as `_batch_evaluate` calls it (src/mcts.py:231-258).
"""
import weakref
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .config import Config
from .engine import Engine
from .game import BreakthroughGame, NUM_ACTIONS, board_to_bits


class _Tree:
    """One engine slot; released when the last Node handle of the tree dies."""

    def __init__(self, pool, engine, slot):
        self.pool = pool
        self.engine = engine
        self.slot = slot
        self.epoch = 0          # bumped whenever node indices change meaning
        self.root_idx = -1      # pool index of the current root, -1 = none yet
        weakref.finalize(self, pool._release, engine, slot)


class Node:
    """MCTS tree node (src/mcts.py:21-40).

    Built by user code (`Node(prior)`) it is a plain host object.  Returned by a
    search it is bound to a device tree: `visit_count`, `value_sum`, `prior` and
    `children` are fetched from the GPU when first read and refreshed by the
    next search on that root.  Handles of nodes that were dropped by a re-root
    raise if read afterwards.
    """

    __slots__ = ["_visit_count", "_value_sum", "_prior", "_children", "_tree", "_idx", "_epoch",
                 "_stats_loaded", "__weakref__"]

    def __init__(self, prior: float):
        self._visit_count = 0
        self._value_sum = 0.0
        self._prior = prior
        self._children: Optional[Dict[int, "Node"]] = {}
        self._tree = None
        self._idx = -1
        self._epoch = 0
        self._stats_loaded = True

    # -- device binding ------------------------------------------------------
    @classmethod
    def _bound(cls, tree, idx, visit_count=None, value_sum=None, prior=0):
        n = cls(prior)
        n._bind(tree, idx)
        if visit_count is not None:
            n._visit_count, n._value_sum, n._stats_loaded = visit_count, value_sum, True
        return n

    def _bind(self, tree, idx):
        self._tree, self._idx, self._epoch = tree, idx, tree.epoch
        self._children = None
        self._stats_loaded = False

    def _invalidate(self):
        self._children = None
        self._stats_loaded = False

    def _query(self):
        t = self._tree
        if self._epoch != t.epoch:
            raise _lib.AzError("this Node belongs to a part of the tree that was dropped by a re-root")
        self._fill(t.engine.node_children(t.slot, self._idx))

    def _fill(self, q):
        t = self._tree
        self._visit_count, self._value_sum, self._stats_loaded = q["root_N"], np.float32(q["root_W"]), True
        ch = {}
        for a, n, w, p, is64, idx in zip(q["actions"], q["N"], q["W"], q["P"], q["p_is64"], q["index"]):
            prior = float(p) if is64 else np.float32(p)
            ch[int(a)] = Node._bound(t, int(idx), int(n), np.float32(w), prior)
        self._children = ch

    # -- reference attributes ---------------------------------------------------
    @property
    def visit_count(self):
        if self._tree is not None and not self._stats_loaded:
            self._query()
        return self._visit_count

    @visit_count.setter
    def visit_count(self, v):
        self._visit_count, self._stats_loaded = v, True

    @property
    def value_sum(self):
        if self._tree is not None and not self._stats_loaded:
            self._query()
        return self._value_sum

    @value_sum.setter
    def value_sum(self, v):
        self._value_sum, self._stats_loaded = v, True

    @property
    def prior(self):
        return self._prior

    @prior.setter
    def prior(self, v):
        self._prior = v

    @property
    def children(self) -> Dict[int, "Node"]:
        if self._children is None:
            self._query()
        return self._children

    def value(self) -> float:
        if self.visit_count == 0:
            return 0.0
        return self.value_sum / self.visit_count

    def is_expanded(self) -> bool:
        return len(self.children) > 0


def _flatten_host_tree(root):
    """Breadth-first arrays of a host-built Node tree for Engine.import_tree, and the Node
    objects in that order.  Types: visit counts int; value sums float32 (what a search leaves
    in them); priors float32, except (i) at the root, where any prior that is not a numpy
    float32 puts all of the root's priors in float64, the state _add_dirichlet_noise leaves
    (src/mcts.py:314-326), and (ii) below the root, where k children that all hold the Python
    float 1/k are the uniform fallback of _expand_node (src/mcts.py:277-278).  Any other plain
    Python float below the root is rounded to float32."""
    nodes, first, count, action = [root], [0], [0], [0]
    i = 0
    while i < len(nodes):
        ch = nodes[i]._children or {}
        if len(ch) > 48:
            raise _lib.AzError(f"a Node with {len(ch)} children: a Breakthrough position has at most 48 moves")
        if ch:
            first[i], count[i] = len(nodes), len(ch)
            for a, c in ch.items():
                if not (0 <= int(a) < NUM_ACTIONS):
                    raise _lib.AzError(f"child key {a!r} is not an action index")
                nodes.append(c)
                first.append(0)
                count.append(0)
                action.append(int(a))
        i += 1
        if len(nodes) > (1 << 17):
            raise _lib.AzError("the host tree has more nodes than an engine slot holds (or refers to itself)")
    uniform = np.zeros(len(nodes), np.uint8)
    for i, nd in enumerate(nodes):
        k = count[i]
        if i > 0 and k and all(type(nodes[first[i] + j]._prior) is float and nodes[first[i] + j]._prior == 1.0 / k
                               for j in range(k)):
            uniform[i] = 1
    root_p64 = None
    k0 = count[0]
    if k0 and not all(isinstance(nodes[1 + j]._prior, np.float32) for j in range(k0)):
        root_p64 = np.array([float(nodes[1 + j]._prior) for j in range(k0)], np.float64)
    arrays = dict(visits=np.array([int(nd._visit_count) for nd in nodes], np.uint32),
                  value_sum=np.array([np.float32(nd._value_sum) for nd in nodes], np.float32),
                  prior=np.array([np.float32(nd._prior) for nd in nodes], np.float32),
                  first_child=np.array(first, np.uint32), n_children=np.array(count, np.uint8),
                  action=np.array(action, np.uint8), uniform64=uniform, root_prior64=root_p64)
    return arrays, nodes


class _SlotPool:
    """Engines are created on demand; each serves up to `capacity` trees."""

    def __init__(self, c_puct, capacity=64, device=None):
        self.c_puct = c_puct
        self.capacity = capacity
        self.device = device
        self.engines: List[Engine] = []
        self.free: Dict[int, List[int]] = {}

    def _release(self, engine, slot):
        self.free[id(engine)].append(slot)

    def acquire(self, want):
        """-> list of (engine, slot), preferring to pack them into one engine."""
        out = []
        for e in self.engines:
            fl = self.free[id(e)]
            while fl and len(out) < want:
                out.append((e, fl.pop()))
        while len(out) < want:
            cap = max(self.capacity, want - len(out))
            e = Engine(cap, c_puct=self.c_puct, dirichlet_alpha=Config.DIRICHLET_ALPHA,
                       dirichlet_epsilon=Config.DIRICHLET_EPSILON,
                       temperature_threshold=Config.TEMPERATURE_THRESHOLD,
                       pool_nodes_per_game=1 << 17, device=self.device)
            e.set_active(np.zeros(cap, np.uint8))
            self.engines.append(e)
            self.free[id(e)] = list(range(cap - 1, -1, -1))
            fl = self.free[id(e)]
            while fl and len(out) < want:
                out.append((e, fl.pop()))
        return out


class MCTS:
    """Drop-in for src/mcts.py:43-342."""

    def __init__(self, model, num_simulations: int = Config.MCTS_SIMULATIONS, c_puct: float = Config.C_PUCT,
                 fpu_reduction: float = Config.FPU_REDUCTION, device: str = "cpu", leaf_batch: int = 1):
        """leaf_batch (no counterpart in the reference): descents of a game per network call.  1 is
        the reference's search, bit for bit.  k > 1 evaluates k leaves of a game at once (virtual
        loss keeps the descents apart): the search of ONE game, as arena and web run it
        (src/arena.py:248-265, src/web.py:447-470), then takes a fraction of the time, but visit
        counts are no longer the reference's.  Needs the HIP network path (an AlphaZeroNet the
        kernels support, on a GPU)."""
        if not 1 <= int(leaf_batch) <= 32:
            raise ValueError("leaf_batch must be in 1..32")
        self.leaf_batch = int(leaf_batch)
        self.model = model
        self.num_simulations = num_simulations
        self.c_puct = c_puct
        self.fpu_reduction = fpu_reduction   # accepted and unused, like the reference (src/mcts.py:296-298)
        self.device = device
        self._pool = None
        self._hip = None
        self._hip_key = None

    # ------------------------------------------------------------------ public API
    def search(self, game: BreakthroughGame, root: Optional[Node] = None, add_noise: bool = False) -> Node:
        """src/mcts.py:76-92"""
        roots = [root] if root is not None else None
        return self.search_batch([game], roots=roots, add_noise=add_noise)[0]

    def search_batch(self, games: List[BreakthroughGame], roots: Optional[List[Optional[Node]]] = None,
                     add_noise: bool = False) -> List[Node]:
        """src/mcts.py:94-189"""
        if not torch.cuda.is_available():
            raise _lib.AzError("MCTS.search_batch needs a GPU: the search runs in libazamd (no CPU fallback)")
        n = len(games)
        if roots is None:
            roots = [None] * n
        else:
            while len(roots) < n:
                roots.append(None)
        if self._pool is None:
            dev = torch.device(self.device) if str(self.device).startswith("cuda") else None
            self._pool = _SlotPool(self.c_puct, device=None if dev is None else dev.index)
        # bind every game to a device tree
        for r in roots:
            if isinstance(r, Node) and r._tree is not None and r._tree.pool is not self._pool:
                raise _lib.AzError("this root belongs to a tree built by another MCTS object: search it with that "
                                   "object, or pass root=None to start a fresh tree")
        fresh = [i for i in range(n) if not (isinstance(roots[i], Node) and roots[i]._tree is not None)]
        got = self._pool.acquire(len(fresh))
        for i, (eng, slot) in zip(fresh, got):
            tree = _Tree(self._pool, eng, slot)
            if roots[i] is None:
                roots[i] = Node(0)
            # a root the caller built by hand (src/mcts.py:117-125 takes any Node): its tree goes
            # up once the position is set
            tree.upload = _flatten_host_tree(roots[i]) if roots[i]._children else None
            roots[i]._bind(tree, -1)
        by_engine: Dict[int, List[int]] = {}
        for i in range(n):
            by_engine.setdefault(id(roots[i]._tree.engine), []).append(i)
        groups = [(roots[idxs[0]]._tree.engine, idxs) for idxs in by_engine.values()]
        try:
            # 1. positions up, re-roots, missing roots expanded - engine by engine
            for engine, idxs in groups:
                self._prepare_on_engine(engine, [games[i] for i in idxs], [roots[i] for i in idxs])
            # 2. one np.random.dirichlet draw per expanded root in ROOT-INDEX order, whatever engine a
            #    root lives on: the draws of the global numpy generator are consumed exactly as
            #    the reference consumes them (src/mcts.py:141-144, 322)
            noise = None
            if add_noise:
                counts = {id(engine): engine.root_counts() for engine, _ in groups}
                noise = {id(engine): np.zeros((engine.n, 48), np.float64) for engine, _ in groups}
                for i in range(n):
                    t = roots[i]._tree
                    k = int(counts[id(t.engine)][t.slot])
                    if k > 0:
                        noise[id(t.engine)][t.slot, :k] = np.random.dirichlet([Config.DIRICHLET_ALPHA] * k)
            # 3. the simulations, engine by engine
            for engine, idxs in groups:
                self._simulate_on_engine(engine, [roots[i] for i in idxs], None if noise is None else noise[id(engine)])
        finally:
            for engine, _ in groups:                 # also after an error: no game stays active
                engine.set_active(np.zeros(engine.n, np.uint8))
        return roots

    def get_action_probs(self, root: Node, temperature: float = 1.0) -> np.ndarray:
        """src/mcts.py:191-225 (host arithmetic on the visit counts, any temperature)"""
        probs = np.zeros(NUM_ACTIONS, dtype=np.float32)
        for action, child in root.children.items():
            probs[action] = child.visit_count
        if temperature == 0:
            best = int(np.argmax(probs))
            probs = np.zeros(NUM_ACTIONS, dtype=np.float32)
            probs[best] = 1.0
        else:
            probs = np.power(probs, 1.0 / temperature)
            total = probs.sum()
            if total > 0:
                probs /= total
            else:
                legal = list(root.children.keys())
                if legal:
                    probs[legal] = 1.0 / len(legal)
        return probs

    # ------------------------------------------------------------------ internals
    def _hip_inference(self, engine):
        """HipInference for self.model if the fused path applies, refreshed when the
        weights changed (tensor version counters move on every in-place update)."""
        from .model import AlphaZeroNet, HipInference
        m = self.model
        if not (isinstance(m, AlphaZeroNet) and m.hip_supported() and str(self.device).startswith("cuda")):
            return None
        key = tuple(t._version for t in list(m.parameters()) + list(m.buffers()))
        if self._hip is None or self._hip.max_batch < engine.n:
            self._hip = HipInference(m, max_batch=max(engine.n, 256 if self.leaf_batch > 1 else 64), device=engine.device.index)
            self._hip_key = key
        elif key != self._hip_key:
            self._hip.refresh()
            self._hip_key = key
        return self._hip

    def _evaluate(self, engine, planes, need):
        """_batch_evaluate (src/mcts.py:231-258) on the rows that need it."""
        rows = torch.nonzero(need, as_tuple=False).flatten()
        if rows.numel() == 0:
            return engine.probs, engine.values
        batch = planes.index_select(0, rows).to(self.device)
        self.model.eval()
        with torch.no_grad():
            policy_logits, wl_logits = self.model(batch)
        policy_probs = torch.softmax(policy_logits, dim=1)
        wl_probs = torch.softmax(wl_logits, dim=1)
        values = wl_probs[:, 0] - wl_probs[:, 1]
        engine.probs.index_copy_(0, rows, policy_probs.to(engine.device, torch.float32))
        engine.values.index_copy_(0, rows, values.to(engine.device, torch.float32))
        return engine.probs, engine.values

    def _prepare_on_engine(self, engine, games, roots):
        """Positions up, reused trees re-rooted at the node passed in, missing roots expanded."""
        slots = np.array([r._tree.slot for r in roots], np.int32)
        wb = [board_to_bits(g.board) for g in games]
        white = np.array([x[0] for x in wb], dtype=np.uint64)
        black = np.array([x[1] for x in wb], dtype=np.uint64)
        turn = np.array([g.turn for g in games], np.int8)
        winner = np.array([g._winner or 0 for g in games], np.int8)
        new = np.array([r._idx < 0 for r in roots])
        if new.any():
            engine.set_positions(slots[new], white[new], black[new], turn[new], winner[new])
            for r in roots:
                t = r._tree
                if r._idx < 0 and getattr(t, "upload", None) is not None:
                    arrays, nodes = t.upload
                    engine.import_tree(t.slot, **arrays)
                    t.upload = None
                    t.root_idx = 0
                    for idx, node in enumerate(nodes):     # the caller's Node objects become handles
                        node._bind(t, idx)
        if (~new).any():
            engine.update_positions(slots[~new], white[~new], black[~new], turn[~new], winner[~new])
        move_slots, move_nodes = [], []       # re-root reused trees at the node that was passed in
        for r in roots:
            t = r._tree
            if r._idx >= 0:
                if r._epoch != t.epoch:
                    raise _lib.AzError("root Node belongs to a part of the tree that was dropped")
                if r._idx != t.root_idx:
                    move_slots.append(t.slot)
                    move_nodes.append(r._idx)
                    t.epoch += 1
                    t.root_idx = 0
                    r._idx, r._epoch = 0, t.epoch
        if move_slots:
            engine.set_roots(move_slots, move_nodes)
        active = np.zeros(engine.n, np.uint8)
        active[slots] = 1
        engine.set_active(active)
        hip = self._hip_inference(engine)
        if hip is not None and engine.net is not hip:
            engine.attach_net(hip)
        if hip is not None:
            engine.search(0, add_noise=False)          # expand missing roots with the bf16 network
        else:
            engine.roots_prepare()
            probs, _ = self._evaluate(engine, engine.planes, engine.need)
            engine.roots_expand(probs)

    def _simulate_on_engine(self, engine, roots, noise):
        slots = np.array([r._tree.slot for r in roots], np.int32)
        active = np.zeros(engine.n, np.uint8)
        active[slots] = 1
        engine.set_active(active)
        hip = self._hip_inference(engine)
        if noise is not None:
            engine.add_noise(noise)
        if self.leaf_batch > 1:
            if hip is None:
                raise _lib.AzError("leaf_batch > 1 needs the HIP network path (a supported AlphaZeroNet on a GPU device)")
            per_call = max(1, 256 // self.leaf_batch)         # games per call: rows = games x leaf_batch <= 256
            for lo in range(0, len(slots), per_call):
                engine.search_leaf_parallel(slots[lo:lo + per_call], self.num_simulations, self.leaf_batch)
        elif hip is not None:
            engine.search(self.num_simulations, add_noise=False)
        else:
            for _ in range(self.num_simulations):
                engine.select()
                probs, values = self._evaluate(engine, engine.planes, engine.need)
                engine.expand_backup(probs, values)
        engine.pool_usage()                   # reports a capacity error of this search
        engine.set_active(np.zeros(engine.n, np.uint8))
        # every returned root is read next (visit counts, children): one launch and one copy
        # for all of them instead of one per handle
        for r, q in zip(roots, engine.roots_children(slots)):
            t = r._tree
            if r._idx < 0:                    # the engine created the root at index 0 of the arena
                r._idx, t.root_idx = 0, 0
            if q is None:
                r._invalidate()
            else:
                r._fill(q)
