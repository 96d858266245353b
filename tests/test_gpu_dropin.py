"""GPU: the drop-in MCTS / Node surface (alphazero_amd.mcts) — the reference's own
test semantics (tests/test_mcts.py) plus bit-exact parity of what the Node handles
report against the reference-generated tree fixtures."""
import os

import numpy as np
import pytest
import torch

from fake_net import TableEvaluator
from test_oracle_golden import _check_children

pytestmark = pytest.mark.gpu


def _game(w, b, turn):
    from alphazero_amd.game import BreakthroughGame, bits_to_board
    return BreakthroughGame(board=bits_to_board(int(w), int(b)), turn=int(turn))


def _table_mcts(sims):
    from alphazero_amd.mcts import MCTS

    class TableMCTS(MCTS):
        """Only the network call is replaced, as in tests/golden/make_golden.py."""
        table = TableEvaluator()

        def _evaluate(self, engine, planes, need):
            p, v = self.table(planes.cpu().numpy())
            engine.probs.copy_(torch.from_numpy(np.ascontiguousarray(p)))
            engine.values.copy_(torch.from_numpy(np.ascontiguousarray(v)))
            return engine.probs, engine.values
    return TableMCTS(None, num_simulations=sims, c_puct=1.5)


def _stats(root):
    acts = list(root.children.keys())
    ch = [root.children[a] for a in acts]
    return dict(actions=np.array(acts, np.uint8), N=np.array([c.visit_count for c in ch], np.uint32),
                W=np.array([c.value_sum for c in ch], np.float32),
                P=np.array([float(c.prior) for c in ch], np.float64),
                p_is64=np.array([int(not isinstance(c.prior, np.float32)) for c in ch], np.uint8),
                root_N=root.visit_count)


@pytest.mark.parametrize("S", [2, 100])
def test_node_handles_report_reference_statistics(golden_dir, S):
    d = np.load(os.path.join(golden_dir, "tree_tier.npz"))
    games = [_game(d["white"][g], d["black"][g], d["turn"][g]) for g in range(len(d["white"]))]
    roots = _table_mcts(S).search_batch(games, None, add_noise=False)
    for g, r in enumerate(roots):
        _check_children(_stats(r), d, f"fresh{S}", g)


def test_tree_reuse_through_node_handles_matches_reference(golden_dir):
    """Pass root.children[a] back as the next root, as parallel_trainer.py:195-198 does."""
    d = np.load(os.path.join(golden_dir, "tree_tier.npz"))
    sel = d["reuse_games"]
    m = _table_mcts(int(d["reuse_sims"]))
    games = [_game(d["white"][g], d["black"][g], d["turn"][g]) for g in sel]
    roots = [None] * len(games)
    alive = list(range(len(games)))
    for ply in range(d["reuse_played"].shape[0]):
        if not alive:
            break
        new = m.search_batch([games[i] for i in alive], [roots[i] for i in alive], add_noise=False)
        for j, i in enumerate(alive):
            roots[i] = new[j]
        for i in list(alive):
            _check_children(_stats(roots[i]), d, "reuse", (ply, i))
            probs = m.get_action_probs(roots[i], temperature=1.0)
            assert np.isclose(probs.sum(), 1.0, atol=1e-5)
            a = int(d["reuse_played"][ply, i])
            games[i].step(games[i].decode_action(a))
            roots[i] = roots[i].children[a] if a in roots[i].children else None
            if games[i].is_terminal():
                alive.remove(i)


def _node_from_arrays(d, c):
    """Plain host Node objects from the breadth-first arrays of handbuilt_tier.npz, with the
    value types the reference held (prior: numpy float32 / numpy float64 / Python float)."""
    from alphazero_amd.mcts import Node
    n = int(d[f"{c}_tree_n"])
    P, pt = d[f"{c}_tree_P"], d[f"{c}_tree_ptype"]
    nodes = []
    for i in range(n):
        prior = np.float32(P[i]) if pt[i] == 0 else (np.float64(P[i]) if pt[i] == 1 else float(P[i]))
        nd = Node(prior)
        nd.visit_count = int(d[f"{c}_tree_N"][i])
        nd.value_sum = np.float32(d[f"{c}_tree_W"][i]) if nd.visit_count else 0.0
        nodes.append(nd)
    for i in range(n):
        f, k = int(d[f"{c}_tree_first"][i]), int(d[f"{c}_tree_count"][i])
        for j in range(f, f + k):
            nodes[i].children[int(d[f"{c}_tree_action"][j])] = nodes[j]
    return nodes[0]


def _check_case(st, d, prefix):
    k = int(d[f"{prefix}_k"])
    assert st["root_N"] == int(d[f"{prefix}_root_N"]) and len(st["actions"]) == k
    assert np.array_equal(st["actions"], d[f"{prefix}_actions"][:k])
    assert np.array_equal(st["N"], d[f"{prefix}_N"][:k])
    assert np.array_equal(st["W"].view(np.uint32), d[f"{prefix}_W"][:k].view(np.uint32))
    assert np.array_equal(st["p_is64"], d[f"{prefix}_p_is64"][:k])
    assert np.array_equal(st["P"].view(np.uint64), d[f"{prefix}_P"][:k].view(np.uint64))


def test_search_from_hand_built_roots_matches_reference(golden_dir):
    """search_batch takes any Node as a root (src/mcts.py:117-125): a tree the caller rebuilt node
    by node from an earlier search (up to 1,909 nodes, uniform-fallback nodes among them) and
    roots with typed-in children (a subset of the legal moves in the caller's order, float64
    priors, preset statistics).  The host tree is uploaded into an engine slot and searched on;
    root and most-visited child must report what the reference reported (handbuilt_tier.npz),
    and the caller's own Node objects must show the new statistics."""
    d = np.load(os.path.join(golden_dir, "handbuilt_tier.npz"))
    for i in range(int(d["n_cases"])):
        c = f"c{i}"
        game = _game(d[f"{c}_white"], d[f"{c}_black"], d[f"{c}_turn"])
        root = _node_from_arrays(d, c)
        held = dict(root.children)                       # the caller's own child objects
        m = _table_mcts(int(d[f"{c}_sims"]))
        out = m.search_batch([game], [root], add_noise=False)[0]
        assert out is root
        _check_case(_stats(out), d, f"{c}_root")
        best = int(d[f"{c}_best"])
        _check_case(_stats(out.children[best]), d, f"{c}_child")
        assert held[best].visit_count == out.children[best].visit_count
    # the same object searched again continues the device tree
    again = m.search_batch([game], [root], add_noise=False)[0]
    assert again.visit_count == int(d[f"{c}_root_root_N"]) + int(d[f"{c}_sims"])


def test_leaf_parallel_search_keeps_the_tree_consistent(golden_dir):
    """MCTS(leaf_batch=k): k descents of a game per network call with virtual loss (the latency
    mode for arena / web style single-game searches).  Not the reference's visit counts by
    construction; what must hold: leaf_batch=1 is the exact search; for k > 1 every tree is a
    consistent search tree (root visits = simulations, child visits sum to them, children are the
    legal moves in insertion order, every in-flight count is back to zero, so a second search on
    the same root adds exactly its simulations), and the chosen move mostly agrees with the exact
    search on the ten roots of latency_tier.npz."""
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.mcts import MCTS
    d = np.load(os.path.join(golden_dir, "latency_tier.npz"))
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=20, num_filters=128).cuda().eval()
    games = [_game(d["white"][g], d["black"][g], d["turn"][g]) for g in range(len(d["white"]))]
    S = 200

    def best(root):
        acts = list(root.children.keys())
        return acts[int(np.argmax([root.children[a].visit_count for a in acts]))]

    exact = [best(MCTS(model, num_simulations=S, device="cuda").search(g.clone())) for g in games]
    for k in (8, 16):
        m = MCTS(model, num_simulations=S, device="cuda", leaf_batch=k)
        agree = 0
        for gi, g in enumerate(games):
            root = m.search(g.clone(), add_noise=False)
            acts = [g.encode_action(mv) for mv in g.get_legal_moves()]
            assert list(root.children.keys()) == acts
            assert root.visit_count == S
            assert sum(c.visit_count for c in root.children.values()) == S
            for c in root.children.values():                 # a child's visits = 1 (its own evaluation) + its children's
                if c.visit_count > 0 and c.is_expanded():
                    assert sum(x.visit_count for x in c.children.values()) == c.visit_count - 1
            agree += best(root) == exact[gi]
            if gi == 0:
                again = m.search(g.clone(), root=root, add_noise=False)
                assert again.visit_count == 2 * S and sum(c.visit_count for c in again.children.values()) == 2 * S
        print(f"leaf_batch {k}: most-visited move equals the exact search's on {agree} of {len(games)} roots")
        assert agree >= 6
    # a batch of games in one call, and more simulations than a multiple of the batch
    m = MCTS(model, num_simulations=37, device="cuda", leaf_batch=8)
    roots = m.search_batch([g.clone() for g in games], None, add_noise=True)
    for r in roots:
        assert r.visit_count == 37 and sum(c.visit_count for c in r.children.values()) == 37


class TestReferenceSemantics:      # tests/test_mcts.py:48-195 with the real (bf16) network
    @pytest.fixture(scope="class")
    def mcts(self):
        from alphazero_amd.mcts import MCTS
        from alphazero_amd.model import AlphaZeroNet
        torch.manual_seed(0)
        model = AlphaZeroNet(num_blocks=2, num_filters=128).cuda().eval()
        return MCTS(model=model, num_simulations=10, c_puct=1.5, fpu_reduction=0.25, device="cuda")

    def test_search_returns_expanded_root(self, mcts):
        from alphazero_amd.game import BreakthroughGame
        from alphazero_amd.mcts import Node
        root = mcts.search(BreakthroughGame())
        assert isinstance(root, Node) and root.is_expanded() and len(root.children) == 22
        assert root.visit_count >= mcts.num_simulations
        assert sum(c.visit_count for c in root.children.values()) == mcts.num_simulations

    def test_dirichlet_noise_changes_priors(self, mcts):
        from alphazero_amd.game import BreakthroughGame
        g = BreakthroughGame()
        p1 = [c.prior for c in mcts.search(g, add_noise=False).children.values()]
        p2 = [c.prior for c in mcts.search(g, add_noise=True).children.values()]
        assert p1 != p2
        assert all(isinstance(p, np.float32) for p in p1) and all(isinstance(p, float) for p in p2)

    def test_action_probs(self, mcts):
        from alphazero_amd.game import BreakthroughGame
        root = mcts.search(BreakthroughGame())
        p = mcts.get_action_probs(root, temperature=1.0)
        assert p.shape == (192,) and p.dtype == np.float32 and np.isclose(p.sum(), 1.0, atol=1e-5)
        p0 = mcts.get_action_probs(root, temperature=0.0)
        assert p0.max() == 1.0 and np.count_nonzero(p0) == 1

    def test_tree_reuse_adds_visits(self, mcts):
        from alphazero_amd.game import BreakthroughGame
        g = BreakthroughGame()
        root = mcts.search(g)
        before = root.visit_count
        root = mcts.search(g, root=root)
        assert root.visit_count == before + mcts.num_simulations

    def test_play_game_segment_with_generic_model(self):
        """A duck-typed model (src/mcts.py:248-250) on the CPU: tree on the GPU, network
        wherever the caller put it."""
        from alphazero_amd.game import BreakthroughGame
        from alphazero_amd.mcts import MCTS
        from alphazero_amd.model import AlphaZeroNet
        torch.manual_seed(1)
        np.random.seed(1)
        m = MCTS(AlphaZeroNet(num_blocks=1, num_filters=16).eval(), num_simulations=12, device="cpu")
        g = BreakthroughGame()
        root = None
        for _ in range(8):
            root = m.search(g, root=root, add_noise=True)
            probs = m.get_action_probs(root, temperature=1.0)
            a = int(np.random.choice(len(probs), p=probs))
            assert probs[a] > 0 and a in root.children
            g.step(g.decode_action(a))
            root = root.children[a]
        assert not g.is_terminal()

    def test_stale_handle_raises(self, mcts):
        from alphazero_amd.game import BreakthroughGame
        from alphazero_amd._lib import AzError
        g = BreakthroughGame()
        root = mcts.search(g)
        a, child = next(iter(root.children.items()))
        sibling = [c for k, c in root.children.items() if k != a][0]
        g.step(g.decode_action(a))
        mcts.search(g, root=child)
        with pytest.raises(AzError):
            sibling.children


def _arena_move(root):
    """src/arena.py:257-262 / src/web.py:455-460, verbatim in meaning: iterate root.children in
    insertion order, keep the first strictly largest visit count."""
    best_action, best_visits = -1, -1
    for action, child in root.children.items():
        if child.visit_count > best_visits:
            best_visits, best_action = child.visit_count, action
    return best_action


def test_latency_search_picks_the_reference_moves(golden_dir):
    """MCTS.search(game, add_noise=False) with a fresh root at 200 simulations, as Arena.play_game
    and make_alphazero_move call it: root statistics, the move picked from them by insertion
    order, and root.value() equal the reference's; the game object is not touched."""
    d = np.load(os.path.join(golden_dir, "latency_tier.npz"))
    m = _table_mcts(int(d["sims"]))
    for g in range(len(d["white"])):
        game = _game(d["white"][g], d["black"][g], d["turn"][g])
        board0, turn0 = game.board.copy(), game.turn
        root = m.search(game, add_noise=False)
        assert np.array_equal(game.board, board0) and game.turn == turn0
        _check_children(_stats(root), d, "fresh", g)
        assert _arena_move(root) == int(d["best_action"][g])
        assert float(root.value()) == float(d["root_value"][g])


def test_arena_game_through_the_dropin_equals_the_reference_game(golden_dir):
    """A whole Arena.play_game (src/arena.py:231-270) through the drop-in classes: same moves,
    same visit counts behind them, same final position and result."""
    from alphazero_amd.game import BreakthroughGame, board_to_bits
    d = np.load(os.path.join(golden_dir, "latency_tier.npz"))
    m = _table_mcts(int(d["sims"]))
    game = BreakthroughGame()
    for ply, want in enumerate(d["game_actions"]):
        assert not game.is_terminal()
        root = m.search(game, add_noise=False)
        a = _arena_move(root)
        assert a == int(want), ply
        assert root.children[a].visit_count == int(d["game_best_visits"][ply])
        game.step(game.decode_action(a))
    assert game.is_terminal()
    assert tuple(np.float32(x) for x in game.get_result()) == tuple(d["game_result"])
    assert tuple(int(x) for x in board_to_bits(game.board)) == tuple(int(x) for x in d["game_final"])


def test_noise_draws_follow_root_order_across_engines(golden_dir, monkeypatch):
    """The reference draws one np.random.dirichlet vector per root in root-index order
    (src/mcts.py:141-144).  With the roots of one search_batch spread over two engines in an
    interleaved pattern the drop-in must still consume the global generator in that order:
    the recorded noisy tree-reuse sequence comes out bit for bit."""
    from alphazero_amd.mcts import Node, _SlotPool, _Tree
    d = np.load(os.path.join(golden_dir, "tree_tier.npz"))
    sel = d["noisy_games"]
    m = _table_mcts(int(d["noisy_sims"]))
    pool = m._pool = _SlotPool(m.c_puct, capacity=2)
    a = pool.acquire(2)                    # engine A, both slots
    b = pool.acquire(2)                    # engine B, both slots
    assert a[0][0] is a[1][0] and b[0][0] is b[1][0] and a[0][0] is not b[0][0]
    placement = [b[0], a[0], b[1], a[1]]   # games 0 and 2 on engine B, games 1 and 3 on engine A
    games = [_game(d["white"][g], d["black"][g], d["turn"][g]) for g in sel]
    roots = []
    for eng, slot in placement:
        r = Node(0)
        r._bind(_Tree(pool, eng, slot), -1)
        roots.append(r)
    draws = []

    def replay(alpha, *args, **kw):
        ply, i = draws[0]
        draws.pop(0)
        return d["noisy_noise"][ply, i][:len(alpha)]
    monkeypatch.setattr(np.random, "dirichlet", replay)
    alive = list(range(len(games)))
    for ply in range(d["noisy_played"].shape[0]):
        if not alive:
            break
        draws[:] = [(ply, i) for i in alive]                 # the order the reference drew them in
        new = m.search_batch([games[i] for i in alive], [roots[i] for i in alive], add_noise=True)
        assert not draws
        for j, i in enumerate(alive):
            roots[i] = new[j]
        for i in list(alive):
            _check_children(_stats(roots[i]), d, "noisy", (ply, i))
            act = int(d["noisy_played"][ply, i])
            games[i].step(games[i].decode_action(act))
            roots[i] = roots[i].children[act] if act in roots[i].children else None
            if games[i].is_terminal():
                alive.remove(i)


def test_a_root_of_another_mcts_object_is_refused():
    from alphazero_amd._lib import AzError
    from alphazero_amd.game import BreakthroughGame
    one, two = _table_mcts(4), _table_mcts(4)
    root = one.search(BreakthroughGame())
    with pytest.raises(AzError, match="another MCTS object"):
        two.search(BreakthroughGame(), root=root)
    assert two.search(BreakthroughGame()).visit_count == 4   # and the object still works
