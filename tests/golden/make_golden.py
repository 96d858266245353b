"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run once, in the development container only (needs the reference checkout at
/root/reference; numpy >= 2 promotion rules are part of the pinned semantics):

    python tests/golden/make_golden.py

Outputs (all data, no code):
    fake_table.npz   table of (policy probs, value) rows for the fake network
    game_tier.npz    positions -> ordered legal actions, masks, encodings, results
    tree_tier.npz    per-root child statistics after S simulations, tree reuse,
                     noisy roots with the recorded Dirichlet draws
    selfplay_tier.npz  execute_parallel_episodes stream with recorded RNG draws
    nn_tier.npz      float32 logits of seeded AlphaZeroNet instances
    latency_tier.npz   MCTS.search(game, add_noise=False) with a fresh root at 200 simulations, the
                     move arena.py / web.py pick from it, and one whole arena-style game
    handbuilt_tier.npz  search_batch on roots built by hand (a rebuilt tree; typed-in children)
    baseline_movegen_tier.npz  legal-move sets and terminal results of the reference's second,
                     independently written move generator (src/baseline/state.py)
"""
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/
sys.path.insert(0, "/root/reference")

from src.game import BreakthroughGame, WHITE, BLACK  # noqa: E402
from src.mcts import MCTS, Node                      # noqa: E402
from src.model import AlphaZeroNet                   # noqa: E402
import src.parallel_trainer as ref_pt                # noqa: E402

import fake_net                                      # noqa: E402


def game_bits(g):
    w = b = 0
    for r in range(8):
        for c in range(8):
            if g.board[r, c] == WHITE:
                w |= 1 << (r * 8 + c)
            elif g.board[r, c] == BLACK:
                b |= 1 << (r * 8 + c)
    return w, b


def game_from_bits(w, b, turn, winner=None):
    board = np.zeros((8, 8), np.int8)
    for sq in range(64):
        if (w >> sq) & 1:
            board[sq // 8, sq % 8] = WHITE
        elif (b >> sq) & 1:
            board[sq // 8, sq % 8] = BLACK
    g = BreakthroughGame(board=board, turn=turn)
    g._winner = winner
    return g


# ------------------------------------------------------------------ table
def make_table(T=128):
    rs = np.random.RandomState(1234)
    logits = (rs.standard_normal((T, 192)) * 2.0).astype(np.float32)
    wl = (rs.standard_normal((T, 2)) * 1.0).astype(np.float32)
    probs = torch.softmax(torch.tensor(logits), dim=1).numpy()
    wlp = torch.softmax(torch.tensor(wl), dim=1).numpy()
    values = wlp[:, 0] - wlp[:, 1]
    # a few degenerate rows: all mass on never-legal actions (168..191 are moves
    # from the far row) so that the legal prior sum underflows to exactly 0 and
    # the uniform fallback of _expand_node (mcts.py:277-278) is exercised
    for r in (5, 77):
        probs[r] = 0
        probs[r, 190] = 1.0
    np.savez(os.path.join(HERE, "fake_table.npz"), probs=probs.astype(np.float32),
             values=values.astype(np.float32))


class TableMCTS(MCTS):
    """Reference MCTS with only the network call replaced by the table."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.table = fake_net.TableEvaluator()

    def _batch_evaluate(self, games):
        if not games:
            return []
        states = np.array([g.get_encoded_state() for g in games])
        probs, values = self.table(states)
        return list(zip(probs, values))   # rows / np.float32 scalars, as mcts.py:258


# ------------------------------------------------------------------ game tier
def make_game_tier():
    rng = random.Random(7)
    rows = []

    def record(g):
        w, b = game_bits(g)
        moves = g.get_legal_moves()
        acts = [g.encode_action(m) for m in moves]
        res = g.get_result()
        rows.append(dict(
            white=w, black=b, turn=g.turn, winner=(g._winner or 0),
            k=len(acts), actions=acts + [255] * (48 - len(acts)),
            moves=[list(m) for m in moves] + [[255] * 4] * (48 - len(moves)),
            mask=np.packbits(g.get_legal_action_mask()),
            planes=np.packbits(g.get_encoded_state() > 0.5),
            terminal=int(g.is_terminal()), w=res[0], l=res[1]))

    for _ in range(40):                       # reachable positions, terminal ones included
        g = BreakthroughGame()
        record(g)
        while not g.is_terminal():
            g.step(rng.choice(g.get_legal_moves()))
            record(g)
    rng2 = random.Random(42)                  # sparse boards as tests/test_compatibility.py:49-78
    for _ in range(20):
        w = b = 0
        for sq in range(64):
            if sq // 8 in (0, 7):
                continue
            p = rng2.random()
            if p < 0.1:
                w |= 1 << sq
            elif p < 0.2:
                b |= 1 << sq
        for turn in (WHITE, BLACK):
            record(game_from_bits(w, b, turn))
    # edge cases: blocked side (no legal moves), single pieces on edges
    record(game_from_bits(1 << 8, 1 << 16, WHITE))         # white a2 blocked by black a3
    record(game_from_bits(1 << 8, 1 << 16, BLACK))
    record(game_from_bits(1 << 15, 1 << 55, WHITE))
    record(game_from_bits(1 << 15, 1 << 55, BLACK))
    out = {k: np.array([r[k] for r in rows]) for k in rows[0] if k not in ("white", "black")}
    out["white"] = np.array([r["white"] for r in rows], dtype=np.uint64)
    out["black"] = np.array([r["black"] for r in rows], dtype=np.uint64)
    for k in ("turn", "winner", "terminal"):
        out[k] = out[k].astype(np.int8)
    for k in ("k", "actions", "moves"):
        out[k] = out[k].astype(np.uint8)
    out["w"] = out["w"].astype(np.float32)
    out["l"] = out["l"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "game_tier.npz"), **out)
    print("game tier:", len(rows), "positions")


# ------------------------------------------------------------------ tree tier
def root_stats(root):
    acts = list(root.children.keys())
    ch = [root.children[a] for a in acts]
    k = len(acts)
    pad = 48 - k
    return dict(
        k=k, root_N=root.visit_count,
        actions=np.array(acts + [255] * pad, np.uint8),
        N=np.array([c.visit_count for c in ch] + [0] * pad, np.uint32),
        W=np.array([np.float32(c.value_sum) for c in ch] + [0] * pad, np.float32),
        P=np.array([float(c.prior) for c in ch] + [0] * pad, np.float64),
        p_is64=np.array([int(not isinstance(c.prior, np.float32)) for c in ch] + [0] * pad, np.uint8))


def tree_positions():
    """Root positions: both colours, opening to late game (terminals in-tree)."""
    rng = random.Random(11)
    out = []
    for plies in (0, 1, 6, 15, 24, 33, 40, 47, 52, 58):
        while True:
            g = BreakthroughGame()
            ok = True
            for _ in range(plies):
                if g.is_terminal():
                    ok = False
                    break
                g.step(rng.choice(g.get_legal_moves()))
            if ok and not g.is_terminal():
                out.append(g)
                break
    return out


def make_tree_tier():
    roots_pos = tree_positions()
    out = {}
    wb = [game_bits(g) for g in roots_pos]
    out["white"] = np.array([x[0] for x in wb], np.uint64)
    out["black"] = np.array([x[1] for x in wb], np.uint64)
    out["turn"] = np.array([g.turn for g in roots_pos], np.int8)
    sims_list = [1, 2, 50, 100, 800]
    out["sims_list"] = np.array(sims_list)
    for S in sims_list:                                   # (i) fresh search, noise off
        m = TableMCTS(None, num_simulations=S, c_puct=1.5)
        roots = m.search_batch([g.clone() for g in roots_pos], None, add_noise=False)
        st = [root_stats(r) for r in roots]
        for key in st[0]:
            out[f"fresh{S}_{key}"] = np.array([s[key] for s in st])
    # (ii) reuse sequence: S=100, 14 plies, actions drawn from the T=1 visit distribution
    for tag, noisy in (("reuse", False), ("noisy", True)):
        S = 100 if not noisy else 60
        plies = 14
        games = [roots_pos[i].clone() for i in (0, 1, 4, 6)]
        nG = len(games)
        m = TableMCTS(None, num_simulations=S, c_puct=1.5)
        roots = [None] * nG
        rs = np.random.RandomState(99)
        actions = np.full((plies, nG), 255, np.int16)
        noise_rec = np.zeros((plies, nG, 48), np.float64)
        stats = []
        alive = [True] * nG
        np.random.seed(2024)
        orig_dirichlet = np.random.dirichlet
        for ply in range(plies):
            idx = [i for i in range(nG) if alive[i]]
            drawn = []

            def rec_dirichlet(alpha, *a, **k):
                v = orig_dirichlet(alpha, *a, **k)
                drawn.append(v)
                return v
            np.random.dirichlet = rec_dirichlet
            try:
                new_roots = m.search_batch([games[i] for i in idx], [roots[i] for i in idx],
                                           add_noise=noisy)
            finally:
                np.random.dirichlet = orig_dirichlet
            for j, i in enumerate(idx):
                roots[i] = new_roots[j]
                if noisy:
                    noise_rec[ply, i, :len(drawn[j])] = drawn[j]
            ply_stats = []
            for i in range(nG):
                if not alive[i]:
                    ply_stats.append(None)
                    continue
                ply_stats.append(root_stats(roots[i]))
                probs = m.get_action_probs(roots[i], temperature=1.0)
                a = int(rs.choice(192, p=probs.astype(np.float64) / probs.astype(np.float64).sum()))
                actions[ply, i] = a
                games[i].step(games[i].decode_action(a))
                roots[i] = roots[i].children[a] if a in roots[i].children else None
                if games[i].is_terminal():
                    alive[i] = False
            stats.append(ply_stats)
        out[f"{tag}_sims"] = np.array(S)
        out[f"{tag}_games"] = np.array([0, 1, 4, 6])
        out[f"{tag}_played"] = actions
        if noisy:
            out[f"{tag}_noise"] = noise_rec
        empty = dict(k=0, root_N=0, actions=np.full(48, 255, np.uint8), N=np.zeros(48, np.uint32),
                     W=np.zeros(48, np.float32), P=np.zeros(48, np.float64), p_is64=np.zeros(48, np.uint8))
        for key in empty:
            out[f"{tag}_{key}"] = np.array([[(s or empty)[key] for s in ps] for ps in stats])
    np.savez_compressed(os.path.join(HERE, "tree_tier.npz"), **out)
    print("tree tier done")


# ------------------------------------------------------------------ wide tree tier
def make_tree_wide_tier():
    """64 root positions from seeded random play at every stage of the game (both colours,
    down to a handful of pieces), one fresh search each without noise (S = 200) and one with
    recorded Dirichlet draws (S = 120)."""
    rng = random.Random(2025)
    roots_pos = []
    while len(roots_pos) < 64:
        g = BreakthroughGame()
        stop = rng.randrange(0, 75)
        for _ in range(stop):
            if g.is_terminal():
                break
            g.step(rng.choice(g.get_legal_moves()))
        if not g.is_terminal():
            roots_pos.append(g)
    out = {}
    wb = [game_bits(g) for g in roots_pos]
    out["white"] = np.array([x[0] for x in wb], np.uint64)
    out["black"] = np.array([x[1] for x in wb], np.uint64)
    out["turn"] = np.array([g.turn for g in roots_pos], np.int8)
    for tag, S, noisy in (("plain", 200, False), ("noise", 120, True)):
        m = TableMCTS(None, num_simulations=S, c_puct=1.5)
        drawn = []
        orig = np.random.dirichlet

        def rec(alpha, *a, **k):
            v = orig(alpha, *a, **k)
            drawn.append(np.concatenate([v, np.zeros(48 - len(v))]))
            return v
        np.random.seed(4242)
        np.random.dirichlet = rec
        try:
            roots = m.search_batch([g.clone() for g in roots_pos], None, add_noise=noisy)
        finally:
            np.random.dirichlet = orig
        st = [root_stats(r) for r in roots]
        for key in st[0]:
            out[f"{tag}_{key}"] = np.array([s[key] for s in st])
        out[f"{tag}_sims"] = np.array(S)
        if noisy:
            assert len(drawn) == len(roots_pos)
            out[f"{tag}_draws"] = np.array(drawn, np.float64)
    np.savez_compressed(os.path.join(HERE, "tree_wide_tier.npz"), **out)
    print("wide tree tier:", len(roots_pos), "positions, turns", int((out["turn"] > 0).sum()), "white to move")


# ------------------------------------------------------------------ exploration-constant tier
CPUCT_CASES = (("c07", 0.7, 150, False), ("c21", 2.1, 150, False), ("c40", 4.0, 150, False),
               ("c21n", 2.1, 90, True))


def make_tree_cpuct_tier():
    """The tree tier's ten roots searched with exploration constants other than 1.5, two of
    them not representable in float32 (0.7, 2.1): pins where the constant is rounded."""
    roots_pos = tree_positions()
    out = {}
    wb = [game_bits(g) for g in roots_pos]
    out["white"] = np.array([x[0] for x in wb], np.uint64)
    out["black"] = np.array([x[1] for x in wb], np.uint64)
    out["turn"] = np.array([g.turn for g in roots_pos], np.int8)
    out["tags"] = np.array([c[0] for c in CPUCT_CASES])
    for tag, c, S, noisy in CPUCT_CASES:
        m = TableMCTS(None, num_simulations=S, c_puct=c)
        drawn = []
        orig = np.random.dirichlet

        def rec(alpha, *a, **k):
            v = orig(alpha, *a, **k)
            drawn.append(np.concatenate([v, np.zeros(48 - len(v))]))
            return v
        np.random.seed(97)
        np.random.dirichlet = rec
        try:
            roots = m.search_batch([g.clone() for g in roots_pos], None, add_noise=noisy)
        finally:
            np.random.dirichlet = orig
        st = [root_stats(r) for r in roots]
        for key in st[0]:
            out[f"{tag}_{key}"] = np.array([s[key] for s in st])
        out[f"{tag}_sims"] = np.array(S)
        out[f"{tag}_c_puct"] = np.array(c, np.float64)
        if noisy:
            out[f"{tag}_draws"] = np.array(drawn, np.float64)
    np.savez_compressed(os.path.join(HERE, "tree_cpuct_tier.npz"), **out)
    print("c_puct tier done")


# ------------------------------------------------------------------ deep reuse tier
def make_tree_deep_tier():
    """Four games played for up to 30 plies at 800 simulations per move with root noise and tree
    reuse: trees of several thousand nodes, long backup chains, reused roots with float64
    noise mixed into float32 priors again and again."""
    roots_pos = tree_positions()
    games = [roots_pos[i].clone() for i in (0, 3, 5, 7)]
    nG, S, plies = len(games), 800, 30
    m = TableMCTS(None, num_simulations=S, c_puct=1.5)
    roots = [None] * nG
    rs = np.random.RandomState(123)
    actions = np.full((plies, nG), 255, np.int16)
    noise_rec = np.zeros((plies, nG, 48), np.float64)
    stats = []
    alive = [True] * nG
    np.random.seed(77)
    orig = np.random.dirichlet
    for ply in range(plies):
        idx = [i for i in range(nG) if alive[i]]
        if not idx:
            stats.append([None] * nG)
            continue
        drawn = []

        def rec(alpha, *a, **k):
            v = orig(alpha, *a, **k)
            drawn.append(v)
            return v
        np.random.dirichlet = rec
        try:
            new_roots = m.search_batch([games[i] for i in idx], [roots[i] for i in idx], add_noise=True)
        finally:
            np.random.dirichlet = orig
        for j, i in enumerate(idx):
            roots[i] = new_roots[j]
            noise_rec[ply, i, :len(drawn[j])] = drawn[j]
        ply_stats = []
        for i in range(nG):
            if not alive[i]:
                ply_stats.append(None)
                continue
            ply_stats.append(root_stats(roots[i]))
            probs = m.get_action_probs(roots[i], temperature=1.0 if ply < 8 else 0.0)
            a = int(rs.choice(192, p=probs.astype(np.float64) / probs.astype(np.float64).sum()))
            actions[ply, i] = a
            games[i].step(games[i].decode_action(a))
            roots[i] = roots[i].children[a] if a in roots[i].children else None
            if games[i].is_terminal():
                alive[i] = False
        stats.append(ply_stats)
    out = {}
    wb = [game_bits(roots_pos[i]) for i in (0, 3, 5, 7)]
    out["white"] = np.array([x[0] for x in wb], np.uint64)
    out["black"] = np.array([x[1] for x in wb], np.uint64)
    out["turn"] = np.array([roots_pos[i].turn for i in (0, 3, 5, 7)], np.int8)
    out["sims"] = np.array(S)
    out["played"] = actions
    out["noise"] = noise_rec
    empty = dict(k=0, root_N=0, actions=np.full(48, 255, np.uint8), N=np.zeros(48, np.uint32),
                 W=np.zeros(48, np.float32), P=np.zeros(48, np.float64), p_is64=np.zeros(48, np.uint8))
    for key in empty:
        out[key] = np.array([[(s or empty)[key] for s in ps] for ps in stats])
    np.savez_compressed(os.path.join(HERE, "tree_deep_tier.npz"), **out)
    print("deep tier: plies played per game", [(actions[:, i] != 255).sum() for i in range(nG)],
          "max root N", int(out["root_N"].max()))


# ------------------------------------------------------------------ action-probability tier
def make_probs_tier():
    """get_action_probs of the reference on hand-built roots: random visit counts (ties, one
    root without any visit, one without children) at temperatures 0, 0.25, 0.5, 1 and 2."""
    rs = np.random.RandomState(8)
    temps = [0.0, 0.25, 0.5, 1.0, 2.0]
    m = MCTS(None, num_simulations=1)
    acts, visits, out = [], [], []
    for case in range(14):
        k = 0 if case == 13 else int(rs.randint(1, 40))
        a = np.sort(rs.choice(168, size=k, replace=False))
        if case == 12:
            n = np.zeros(k, np.int64)
        elif case % 3 == 0:
            n = rs.randint(0, 4, size=k) * 50           # many ties
        else:
            n = rs.randint(0, 800, size=k)
        root = Node(0)
        for ai, ni in zip(a, n):
            c = Node(0.1)
            c.visit_count = int(ni)
            root.children[int(ai)] = c
        acts.append(np.concatenate([a, np.full(48 - k, 255)]).astype(np.uint8))
        visits.append(np.concatenate([n, np.zeros(48 - k, np.int64)]).astype(np.uint32))
        out.append(np.stack([m.get_action_probs(root, temperature=t) for t in temps]))
    np.savez_compressed(os.path.join(HERE, "probs_tier.npz"), temps=np.array(temps), actions=np.array(acts),
                        visits=np.array(visits), probs=np.array(out, np.float32))
    print("probs tier:", len(out), "roots x", len(temps), "temperatures")


# ------------------------------------------------------------------ self-play tier
def make_selfplay_tier(nG=3, S=30, seed=31337, name="selfplay_tier.npz"):
    tmp = tempfile.mkdtemp()
    model = AlphaZeroNet(num_blocks=1, num_filters=8)
    tr = ref_pt.ParallelTrainer(model, device="cpu", num_parallel_games=nG, num_simulations=S,
                                checkpoint_dir=tmp)
    tr.mcts = TableMCTS(None, num_simulations=S, c_puct=1.5)
    dir_draws, choice_draws = [], []
    od, oc = np.random.dirichlet, np.random.choice

    def rd(alpha, *a, **k):
        v = od(alpha, *a, **k)
        dir_draws.append(np.concatenate([v, np.zeros(48 - len(v))]))
        return v

    def rc(n, *a, **k):
        v = oc(n, *a, **k)
        choice_draws.append(int(v))
        return v
    np.random.seed(seed)
    np.random.dirichlet, np.random.choice = rd, rc
    try:
        examples = tr.execute_parallel_episodes(nG)
    finally:
        np.random.dirichlet, np.random.choice = od, oc
    np.savez_compressed(
        os.path.join(HERE, name),
        n_games=np.array(nG), sims=np.array(S),
        dirichlet=np.array(dir_draws, np.float64), choices=np.array(choice_draws, np.int16),
        states=np.packbits(np.array([e[0] for e in examples]) > 0.5, axis=None),
        n_examples=np.array(len(examples)),
        policies=np.array([e[1] for e in examples], np.float32),
        wls=np.array([e[2] for e in examples], np.float32))
    print(name, ":", len(examples), "examples,", len(choice_draws), "moves")


def make_selfplay_wide_tier():
    """16 complete games at 24 simulations: more finishing orders, more temperature-0 plies."""
    make_selfplay_tier(nG=16, S=24, seed=777, name="selfplay_wide_tier.npz")


# ------------------------------------------------------------------ NN tier
def randomize_bn(model, seed):
    """Give every BatchNorm non-trivial statistics, deterministically."""
    g = torch.Generator().manual_seed(seed)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            n = mod.num_features
            mod.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(n, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(n, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(n, generator=g) * 0.1)


def make_nn_tier():
    d = np.load(os.path.join(HERE, "game_tier.npz"))
    sel = np.arange(0, len(d["white"]), max(1, len(d["white"]) // 24))[:24]
    planes = np.unpackbits(d["planes"][sel], axis=1)[:, :192].reshape(-1, 3, 8, 8).astype(np.float32)
    out = dict(sel=sel, planes=np.packbits(planes > 0.5, axis=None))
    for tag, blocks, filters in (("small", 2, 32), ("full", 20, 128)):
        torch.manual_seed(0)
        net = AlphaZeroNet(num_blocks=blocks, num_filters=filters)
        randomize_bn(net, 1)
        net.eval()
        with torch.no_grad():
            pl, wl = net(torch.tensor(planes))
        out[f"{tag}_policy_logits"] = pl.numpy()
        out[f"{tag}_wl_logits"] = wl.numpy()
    np.savez_compressed(os.path.join(HERE, "nn_tier.npz"), **out)
    print("nn tier done")


# ------------------------------------------------------------------ training tier
def train_examples(n=48, seed=5):
    """Synthetic (state, policy, wl) tuples of the shapes the trainer stores."""
    rs = np.random.RandomState(seed)
    ex = []
    for _ in range(n):
        st = (rs.rand(3, 8, 8) > 0.6).astype(np.float32)
        st[2] = 1.0
        pol = rs.rand(192).astype(np.float32) ** 4
        pol /= pol.sum()
        wl = np.array([1.0, 0.0] if rs.rand() < 0.5 else [0.0, 1.0], np.float32)
        ex.append((st, pol, wl))
    return ex


def structure(obj):
    """JSON-able description of a checkpoint value: tensors become shape/dtype."""
    if isinstance(obj, torch.Tensor):
        return {"tensor": list(obj.shape), "dtype": str(obj.dtype)}
    if isinstance(obj, dict):
        return {str(k): structure(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [structure(v) for v in obj]
    return obj


def make_train_tier():
    import json
    out = {}
    ex = train_examples()
    # mirror augmentation (src/parallel_trainer.py:33-70)
    aug = [ref_pt.augment_example(*e) for e in ex[:6]]
    out["aug_states"] = np.array([a[0] for a in aug])
    out["aug_policies"] = np.array([a[1] for a in aug])
    out["aug_wls"] = np.array([a[2] for a in aug])
    # learn(): 2 epochs x 6 batches of 16 over 96 augmented examples, then the scheduler step
    tmp = tempfile.mkdtemp()
    torch.manual_seed(123)
    model = AlphaZeroNet(num_blocks=1, num_filters=8)
    tr = ref_pt.ParallelTrainer(model, device="cpu", num_parallel_games=2, num_simulations=4, checkpoint_dir=tmp)
    tr.add_examples(ex)
    torch.manual_seed(7)
    tr.learn(epochs=2, batch_size=16)
    sd = model.state_dict()
    out["learn_keys"] = np.array(list(sd.keys()))
    out["learn_sums"] = np.array([float(v.double().sum()) for v in sd.values()], np.float64)
    out["learn_abs_sums"] = np.array([float(v.double().abs().sum()) for v in sd.values()], np.float64)
    out["learn_input_conv"] = sd["input_conv.weight"].numpy()
    out["learn_value_fc2"] = sd["value_fc2.weight"].numpy()
    out["learn_lr_after"] = np.array(tr.optimizer.param_groups[0]["lr"], np.float64)
    # on-disk formats: iteration_N.pt (:290-311) and training_data.npz (:367-396)
    tr.save_iteration_checkpoint(3)
    ck = torch.load(os.path.join(tmp, "iteration_3.pt"), weights_only=True)
    out["checkpoint_structure"] = np.array(json.dumps(structure(ck), sort_keys=True))
    tr.append_training_data(ex[:10])
    tr.append_training_data(ex[10:25])
    d = np.load(os.path.join(tmp, "training_data.npz"))
    out["data_keys"] = np.array(sorted(d.files))
    out["data_shapes"] = np.array(json.dumps({k: [list(d[k].shape), str(d[k].dtype)] for k in d.files}, sort_keys=True))
    tr.max_examples = 12
    n = tr.load_training_data()
    out["data_loaded"] = np.array(n)
    out["data_loaded_first_policy"] = tr.examples[0][1]
    tr.add_examples(ex[25:30])
    out["buffer_len_after_add"] = np.array(len(tr.examples))
    out["buffer_first_policy_after_add"] = tr.examples[0][1]
    out["latest_iteration"] = np.array(tr.get_latest_iteration())
    np.savez_compressed(os.path.join(HERE, "train_tier.npz"), **out)
    print("train tier done")


# ------------------------------------------------------------------ latency (arena / web) tier
def best_by_visits(root):
    """The move rule of src/arena.py:257-262 and src/web.py:455-460: the first child in
    insertion order with the strictly largest visit count."""
    best_action, best_visits = -1, -1
    for action, child in root.children.items():
        if child.visit_count > best_visits:
            best_visits = child.visit_count
            best_action = action
    return best_action


def make_latency_tier(S=200):
    """Arena.play_game (src/arena.py:248-265) and make_alphazero_move (src/web.py:447-470) call
    MCTS.search(game, add_noise=False) with a fresh root for every move and play the most
    visited child.  Ten roots searched that way, and one whole game played that way."""
    roots_pos = tree_positions()
    m = TableMCTS(None, num_simulations=S, c_puct=1.5)
    out = {"sims": np.array(S)}
    wb = [game_bits(g) for g in roots_pos]
    out["white"] = np.array([x[0] for x in wb], np.uint64)
    out["black"] = np.array([x[1] for x in wb], np.uint64)
    out["turn"] = np.array([g.turn for g in roots_pos], np.int8)
    st, best, val = [], [], []
    for g in roots_pos:
        before = (g.board.copy(), g.turn)
        root = m.search(g, add_noise=False)
        assert np.array_equal(before[0], g.board) and before[1] == g.turn      # the game is not touched
        st.append(root_stats(root))
        best.append(best_by_visits(root))
        val.append(float(root.value()))                                      # web.py:463
    for key in st[0]:
        out[f"fresh_{key}"] = np.array([s[key] for s in st])
    out["best_action"] = np.array(best, np.int16)
    out["root_value"] = np.array(val, np.float64)
    # one arena game: both sides the same table network, fresh root per move
    g = BreakthroughGame()
    played, visits = [], []
    while not g.is_terminal():
        root = m.search(g, add_noise=False)
        a = best_by_visits(root)
        played.append(a)
        visits.append(root.children[a].visit_count)
        g.step(g.decode_action(a))
    w, l = g.get_result()
    out["game_actions"] = np.array(played, np.int16)
    out["game_best_visits"] = np.array(visits, np.uint32)
    out["game_result"] = np.array([w, l], np.float32)
    fw, fb = game_bits(g)
    out["game_final"] = np.array([fw, fb], np.uint64)
    np.savez_compressed(os.path.join(HERE, "latency_tier.npz"), **out)
    print("latency tier:", len(roots_pos), "roots; arena game of", len(played), "plies, result", (w, l))


# ------------------------------------------------------------------ second movegen oracle
def make_baseline_movegen_tier():
    """src/baseline/state.py is a second move generator, written independently of src/game.py
    on bitboards (shifts by 7 / 8 / 9 with file masks); the reference's own
    tests/test_compatibility.py:40-104 cross-checks the two.  Its move sets (from-square,
    to-square on the real board), is_terminal and check_win for: the start position, the 20
    sparse boards of that test for both sides, 400 positions from seeded random play, and
    200 random boards of varying density with pieces on every row."""
    from src.baseline.state import BitboardState
    rows = []

    def record(w, b, turn):
        stt = BitboardState(white=w, black=b, turn=turn)
        mv = sorted(stt.get_legal_moves())
        assert len(mv) <= 48
        win = stt.check_win()
        rows.append(dict(white=w, black=b, turn=turn, k=len(mv),
                         frm=[m[0] for m in mv] + [255] * (48 - len(mv)),
                         to=[m[1] for m in mv] + [255] * (48 - len(mv)),
                         terminal=int(stt.is_terminal()), win=0 if win is None else int(win)))

    record(0xFFFF, 0xFFFF << 48, WHITE)
    rng = random.Random(42)                    # tests/test_compatibility.py:49-78
    for _ in range(20):
        w = b = 0
        for sq in range(64):
            if sq // 8 in (0, 7):
                continue
            p = rng.random()
            if p < 0.1:
                w |= 1 << sq
            elif p < 0.2:
                b |= 1 << sq
        for turn in (WHITE, BLACK):
            record(w, b, turn)
    rng = random.Random(909)                   # reachable positions, terminal ones included
    while len(rows) < 441:
        g = BreakthroughGame()
        while True:
            w, b = game_bits(g)
            if rng.random() < 0.35:
                record(w, b, g.turn)
            if g.is_terminal():
                break
            g.step(rng.choice(g.get_legal_moves()))
    rng = random.Random(4711)                  # arbitrary boards: any row, any density
    for i in range(200):
        dens = rng.choice((0.05, 0.15, 0.3, 0.45))
        w = b = 0
        for sq in range(64):
            p = rng.random()
            if p < dens:
                w |= 1 << sq
            elif p < 2 * dens:
                b |= 1 << sq
        record(w, b, WHITE if i % 2 == 0 else BLACK)
    out = {"white": np.array([r["white"] for r in rows], np.uint64),
           "black": np.array([r["black"] for r in rows], np.uint64),
           "turn": np.array([r["turn"] for r in rows], np.int8),
           "k": np.array([r["k"] for r in rows], np.uint8),
           "frm": np.array([r["frm"] for r in rows], np.uint8),
           "to": np.array([r["to"] for r in rows], np.uint8),
           "terminal": np.array([r["terminal"] for r in rows], np.int8),
           "win": np.array([r["win"] for r in rows], np.int8)}
    np.savez_compressed(os.path.join(HERE, "baseline_movegen_tier.npz"), **out)
    print("baseline movegen tier:", len(rows), "positions,", int(out["terminal"].sum()), "terminal")

# ------------------------------------------------------------------ hand-built roots
def flatten_tree(root):
    """Breadth-first arrays of a Node tree (children of a node contiguous, insertion order)."""
    nodes, first, count, action = [root], [0], [0], [0]
    i = 0
    while i < len(nodes):
        ch = nodes[i].children
        if ch:
            first[i], count[i] = len(nodes), len(ch)
            for a, c in ch.items():
                nodes.append(c)
                first.append(0)
                count.append(0)
                action.append(a)
        i += 1
    return dict(
        n=len(nodes), first=np.array(first, np.uint32), count=np.array(count, np.uint8), action=np.array(action, np.uint8),
        N=np.array([nd.visit_count for nd in nodes], np.uint32),
        W=np.array([float(nd.value_sum) for nd in nodes], np.float64),
        P=np.array([float(nd.prior) for nd in nodes], np.float64),
        # 0: numpy float32, 1: numpy float64, 2: plain Python float
        ptype=np.array([0 if isinstance(nd.prior, np.float32) else (1 if isinstance(nd.prior, np.float64) else 2)
                        for nd in nodes], np.uint8))


def copy_tree(node):
    """The same tree in freshly constructed Node objects (what a caller that rebuilds a saved tree holds)."""
    c = Node(node.prior)
    c.visit_count, c.value_sum = node.visit_count, node.value_sum
    for a, ch in node.children.items():
        c.children[a] = copy_tree(ch)
    return c


def make_handbuilt_tier():
    """search_batch with roots the caller built by hand (src/mcts.py:117-125 accepts any Node):
    (a) a searched tree rebuilt node by node and searched on; (b) a root whose children were
    typed in: a subset of the legal moves in an order of the caller's choosing, float64 priors,
    preset visit counts and value sums."""
    pos = tree_positions()
    cases = []
    for gi, S1, S2 in ((0, 40, 60), (3, 25, 100), (5, 80, 30), (7, 10, 200)):
        g = pos[gi]
        m = TableMCTS(None, num_simulations=S1, c_puct=1.5)
        root = m.search_batch([g.clone()], None, add_noise=False)[0]
        rebuilt = copy_tree(root)
        tree = flatten_tree(rebuilt)
        m2 = TableMCTS(None, num_simulations=S2, c_puct=1.5)
        out = m2.search_batch([g.clone()], [rebuilt], add_noise=False)[0]
        best = max(out.children, key=lambda a: out.children[a].visit_count)
        cases.append((g, tree, S2, root_stats(out), best, root_stats(out.children[best])))
    rng = random.Random(5)
    for gi, S2 in ((1, 50), (4, 120), (6, 64)):
        g = pos[gi]
        acts = [g.encode_action(mv) for mv in g.get_legal_moves()]
        rng.shuffle(acts)
        acts = acts[: max(2, len(acts) // 2)]
        wts = np.array([rng.random() + 0.1 for _ in acts])
        pri = wts / wts.sum()                              # numpy float64 priors
        root = Node(0)
        root.visit_count = 7
        for a, p_ in zip(acts, pri):
            ch = Node(p_)
            ch.visit_count = rng.choice((0, 0, 1, 3))
            ch.value_sum = np.float32(rng.uniform(-1, 1) * ch.visit_count)
            root.children[a] = ch
        tree = flatten_tree(root)
        m2 = TableMCTS(None, num_simulations=S2, c_puct=1.5)
        out = m2.search_batch([g.clone()], [root], add_noise=False)[0]
        best = max(out.children, key=lambda a: out.children[a].visit_count)
        cases.append((g, tree, S2, root_stats(out), best, root_stats(out.children[best])))
    o = {"n_cases": np.array(len(cases))}
    for i, (g, tree, S2, st, best, st_best) in enumerate(cases):
        w, b = game_bits(g)
        o[f"c{i}_white"], o[f"c{i}_black"], o[f"c{i}_turn"] = np.uint64(w), np.uint64(b), np.int8(g.turn)
        o[f"c{i}_sims"], o[f"c{i}_best"] = np.array(S2), np.array(best)
        for k_, v in tree.items():
            o[f"c{i}_tree_{k_}"] = np.asarray(v)
        for k_, v in st.items():
            o[f"c{i}_root_{k_}"] = np.asarray(v)
        for k_, v in st_best.items():
            o[f"c{i}_child_{k_}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "handbuilt_tier.npz"), **o)
    print("hand-built tier:", len(cases), "cases,", [c[1]["n"] for c in cases], "nodes")


if __name__ == "__main__":
    torch.set_num_threads(4)
    tiers = {"table": make_table, "game": make_game_tier, "tree": make_tree_tier,
             "tree_wide": make_tree_wide_tier, "tree_cpuct": make_tree_cpuct_tier, "tree_deep": make_tree_deep_tier, "probs": make_probs_tier,
             "selfplay": make_selfplay_tier, "selfplay_wide": make_selfplay_wide_tier, "nn": make_nn_tier, "train": make_train_tier,
             "latency": make_latency_tier, "baseline_movegen": make_baseline_movegen_tier, "handbuilt": make_handbuilt_tier}
    for name in (sys.argv[1:] or list(tiers)):
        tiers[name]()
