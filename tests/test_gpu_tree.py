"""GPU parity, tree tier: visit counts, value sums and priors of the device
search must be bit-identical to the reference (golden fixtures) and to the CPU
oracle, given the same network outputs."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from fake_net import TableEvaluator
from test_oracle_golden import _check_children

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tree_tier(golden_dir):
    return np.load(os.path.join(golden_dir, "tree_tier.npz"))


@pytest.mark.parametrize("S", [1, 2, 50, 100, 800])
def test_fresh_search_matches_reference(tree_tier, S):
    from gpu_backend import EngineBackend
    d = tree_tier
    nG = len(d["white"])
    m = EngineBackend(nG, pool_nodes_per_game=1 << 16)
    for g in range(nG):
        m.set_game(g, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
    m.search(TableEvaluator(), S)
    for g in range(nG):
        _check_children(m.root_children(g), d, f"fresh{S}", g)


@pytest.mark.parametrize("tag", ["reuse", "noisy"])
def test_reuse_sequence_matches_reference(tree_tier, tag):
    from gpu_backend import EngineBackend
    d = tree_tier
    games = d[f"{tag}_games"]
    S = int(d[f"{tag}_sims"])
    actions = d[f"{tag}_played"]
    nG = len(games)
    ev = TableEvaluator()
    m = EngineBackend(nG, pool_nodes_per_game=1 << 16)
    for j, g in enumerate(games):
        m.set_game(j, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
    alive = np.ones(nG, bool)
    for ply in range(actions.shape[0]):
        if not alive.any():
            break
        m.set_active(alive)
        noise = d[f"{tag}_noise"][ply] if tag == "noisy" else None
        m.search(ev, S, noise=noise)
        for j in range(nG):
            if not alive[j]:
                continue
            _check_children(m.root_children(j), d, tag, (ply, j))
            m.advance(j, int(actions[ply, j]))
            if orc.is_terminal(m.get_game(j)):
                alive[j] = False


@pytest.mark.parametrize("fixture", ["selfplay_tier.npz", "selfplay_wide_tier.npz"])
def test_selfplay_stream_matches_reference(golden_dir, fixture):
    from gpu_backend import EngineBackend
    from selfplay_replay import replay_selfplay
    d = np.load(os.path.join(golden_dir, fixture))
    nG, S = int(d["n_games"]), int(d["sims"])
    states, policies, wls = replay_selfplay(EngineBackend(nG, pool_nodes_per_game=1 << 16), nG, S,
                                            d["dirichlet"], d["choices"], TableEvaluator())
    n = int(d["n_examples"])
    ref_states = np.unpackbits(d["states"])[: n * 192].reshape(n, 3, 8, 8).astype(np.float32)
    assert np.array_equal(np.array(states), ref_states)
    assert np.array_equal(np.array(policies).view(np.uint32), d["policies"].view(np.uint32))
    assert np.array_equal(np.array(wls), d["wls"])


def test_many_games_vs_oracle():
    """64 games from random mid-game positions, 200 simulations, two plies with
    tree reuse and host-drawn noise: every root statistic equals the oracle's."""
    from gpu_backend import EngineBackend
    nG, S = 64, 200
    w, b, t, win = orc.random_positions(4000, seed=5)
    live = [i for i in range(4000) if win[i] == 0 and orc.legal_actions(orc.make_pos(w[i], b[i], t[i])).size][:: 4000 // nG][:nG]
    ev = TableEvaluator()
    dev, cpu = EngineBackend(nG, pool_nodes_per_game=1 << 16), orc.OracleMCTS(nG)
    for j, i in enumerate(live):
        p = orc.make_pos(w[i], b[i], t[i])
        dev.set_game(j, p); cpu.set_game(j, p)
    rs = np.random.RandomState(0)
    alive = np.ones(nG, bool)
    for ply in range(3):
        dev.set_active(alive); cpu.set_active(alive)
        noise = rs.dirichlet([0.35] * 48, size=nG)
        dev.search(ev, S, noise=noise); cpu.search(ev, S, noise=noise)
        for j in range(nG):
            if not alive[j]:
                continue
            a, o = dev.root_children(j), cpu.root_children(j)
            assert np.array_equal(a["actions"], o["actions"])
            assert np.array_equal(a["N"], o["N"])
            assert np.array_equal(a["W"].view(np.uint32), o["W"].view(np.uint32))
            assert np.array_equal(a["P"].view(np.uint64), o["P"].view(np.uint64))
            assert a["root_N"] == o["root_N"]
            assert np.array_equal(dev.action_probs(j, 1.0).view(np.uint32), cpu.action_probs(j, 1.0).view(np.uint32))
            assert np.array_equal(dev.action_probs(j, 0.0), cpu.action_probs(j, 0.0))
            act = int(o["actions"][np.argmax(o["N"])])
            dev.advance(j, act); cpu.advance(j, act)
            if orc.is_terminal(cpu.get_game(j)):
                alive[j] = False
    assert abs(dev.mean_depth() - cpu.mean_depth()) < 1e-9


def test_batched_root_queries_and_reroots_equal_single_calls():
    """az_engine_roots_children / az_engine_set_roots against their one-game forms."""
    from alphazero_amd._lib import AzError
    from gpu_backend import EngineBackend
    nG, S = 12, 60
    ev = TableEvaluator()
    a, b = EngineBackend(nG, pool_nodes_per_game=1 << 14), EngineBackend(nG, pool_nodes_per_game=1 << 14)
    for be in (a, b):
        for g in range(nG):
            be.set_game(g, orc.start_position())
        be.set_active(np.ones(nG, bool))
        be.search(ev, S)
    ea, eb = a.e, b.e
    slots = np.array([7, 0, 11, 3], np.int32)
    for q, g in zip(ea.roots_children(slots), slots):
        one = ea.node_children(int(g))
        for key in ("actions", "N", "W", "P", "p_is64", "index"):
            assert np.array_equal(q[key], one[key]), key
        assert (q["root_N"], q["root_W"], q["node"]) == (one["root_N"], one["root_W"], one["node"])
    # re-root four games at their most visited child: batched on a, one by one on b
    nodes = [int(ea.node_children(int(g))["index"][np.argmax(ea.node_children(int(g))["N"])]) for g in slots]
    ea.set_roots(slots, nodes)
    for g, nd in zip(slots, nodes):
        eb.set_root(int(g), nd)
    for g in range(nG):
        qa, qb = ea.node_children(g), eb.node_children(g)
        for key in ("actions", "N", "W", "P", "index"):
            assert np.array_equal(qa[key], qb[key]), (g, key)
        assert qa["root_N"] == qb["root_N"]
    assert np.array_equal(ea.pool_usage(), eb.pool_usage())
    # dropping a tree, and a node index outside the tree
    ea.set_roots([5], [-1])
    assert ea.roots_children([5])[0] is None
    ea.set_roots([2], [int(ea.pool_usage()[2]) + 5])
    with pytest.raises(AzError):
        ea.pool_usage()


@pytest.mark.parametrize("tag", ["plain", "noise"])
def test_wide_positions_match_reference(golden_dir, tag):
    """64 roots from every stage of the game, searched once by the device engine: children,
    visits, value sums and priors equal the reference's (tree_wide_tier.npz)."""
    from gpu_backend import EngineBackend
    d = np.load(os.path.join(golden_dir, "tree_wide_tier.npz"))
    nG = len(d["white"])
    m = EngineBackend(nG, pool_nodes_per_game=1 << 14)
    for g in range(nG):
        m.set_game(g, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
    m.set_active(np.ones(nG, np.uint8))
    noise = d["noise_draws"] if tag == "noise" else None
    m.search(TableEvaluator(), int(d[f"{tag}_sims"]), noise=noise)
    for g in range(nG):
        _check_children(m.root_children(g), d, tag, g)


@pytest.mark.parametrize("tag", ["c07", "c21", "c40", "c21n"])
def test_other_exploration_constants_match_reference(golden_dir, tag):
    """Exploration constants 0.7, 2.1 and 4.0 on the device engine (tree_cpuct_tier.npz)."""
    from gpu_backend import EngineBackend
    d = np.load(os.path.join(golden_dir, "tree_cpuct_tier.npz"))
    nG = len(d["white"])
    m = EngineBackend(nG, c_puct=float(d[f"{tag}_c_puct"]), pool_nodes_per_game=1 << 14)
    for g in range(nG):
        m.set_game(g, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
    m.set_active(np.ones(nG, np.uint8))
    noise = d[f"{tag}_draws"] if f"{tag}_draws" in d else None
    m.search(TableEvaluator(), int(d[f"{tag}_sims"]), noise=noise)
    for g in range(nG):
        _check_children(m.root_children(g), d, tag, g)


def test_deep_reuse_matches_reference(golden_dir):
    """800 simulations per move, recorded root noise, tree reuse over up to 30 plies
    (tree_deep_tier.npz) on the device engine."""
    from gpu_backend import EngineBackend
    from test_oracle_golden import replay_deep_tier
    d = np.load(os.path.join(golden_dir, "tree_deep_tier.npz"))
    checked, top = replay_deep_tier(EngineBackend(len(d["white"]), pool_nodes_per_game=1 << 16), d)
    assert checked == int((d["played"] != 255).sum()) and top > 1500


def test_action_probs_at_any_temperature(golden_dir):
    """get_action_probs (src/mcts.py:191-225) on the device for the temperatures of
    tests/golden/probs_tier (0, 0.25, 0.5, 1, 2) on searched roots: 0 and 1 bit for bit,
    the others against the reference's own formula evaluated with numpy (np.power, sum,
    divide in float32) within relative 1e-5."""
    from fake_net import TableEvaluator
    from gpu_backend import EngineBackend
    d = np.load(os.path.join(golden_dir, "tree_wide_tier.npz"))
    nG = len(d["white"])
    dev = EngineBackend(nG, pool_nodes_per_game=1 << 14)
    for g in range(nG):
        dev.set_game(g, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
    dev.search(TableEvaluator(), 200)
    temps_cycle = [0.0, 0.25, 0.5, 1.0, 2.0]
    temps = np.array([temps_cycle[g % 5] for g in range(nG)], np.float32)
    got = dev.e.action_probs_all(temps).cpu().numpy()
    for g in range(nG):
        q = dev.root_children(g)
        assert np.array_equal(q["N"], d["plain_N"][g][: len(q["N"])])
        want = np.zeros(192, np.float32)
        want[q["actions"]] = q["N"]
        T = float(temps[g])
        if T == 0:
            best = int(np.argmax(want))
            want = np.zeros(192, np.float32)
            want[best] = 1.0
        else:
            want = np.power(want, 1.0 / T)
            want /= want.sum()
        if T in (0.0, 1.0):
            assert np.array_equal(got[g].view(np.uint32), want.view(np.uint32)), (g, T)
        else:
            np.testing.assert_allclose(got[g], want, rtol=1e-5, atol=1e-9)
            assert np.count_nonzero(got[g]) == np.count_nonzero(want)
