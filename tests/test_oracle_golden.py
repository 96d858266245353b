"""Pins the CPU oracle (oracle/az_oracle.c) to the fixtures that were generated
by running the reference itself (tests/golden/make_golden.py).  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from fake_net import TableEvaluator


@pytest.fixture(scope="module")
def game_tier(golden_dir):
    return np.load(os.path.join(golden_dir, "game_tier.npz"))


@pytest.fixture(scope="module")
def tree_tier(golden_dir):
    return np.load(os.path.join(golden_dir, "tree_tier.npz"))


def test_game_tier_bit_exact(game_tier):
    d = game_tier
    n = len(d["white"])
    assert n > 2000
    for i in range(n):
        p = orc.make_pos(d["white"][i], d["black"][i], d["turn"][i], d["winner"][i])
        acts = orc.legal_actions(p)
        k = int(d["k"][i])
        assert len(acts) == k
        assert np.array_equal(acts, d["actions"][i][:k])           # order included
        assert np.array_equal(np.packbits(orc.legal_mask(p)), d["mask"][i])
        assert np.array_equal(np.packbits(orc.encode_state(p) > 0.5), d["planes"][i])
        assert orc.is_terminal(p) == bool(d["terminal"][i])
        assert orc.get_result(p) == (float(d["w"][i]), float(d["l"][i]))


def test_step_reproduces_reference_games(game_tier):
    """Replaying consecutive fixture positions through azo_step_action must give
    the reference's next position, winner included (game.py:156-187)."""
    d = game_tier
    checked = 0
    for i in range(len(d["white"]) - 1):
        if d["terminal"][i]:
            continue
        p = orc.make_pos(d["white"][i], d["black"][i], d["turn"][i], d["winner"][i])
        nxt = (d["white"][i + 1], d["black"][i + 1], d["turn"][i + 1], d["winner"][i + 1])
        if d["turn"][i + 1] != -d["turn"][i]:
            continue                                   # fixture boundary, not a game step
        found = False
        for a in orc.legal_actions(p):
            q = orc.step_action(p, a)
            if (q.white, q.black, q.turn, q.winner) == tuple(int(x) for x in nxt):
                found = True
                break
        if found:
            checked += 1
    assert checked > 1500


def test_batch_status_matches_winner(game_tier):
    d = game_tier
    legal3, mask, planes, status = orc.movegen_batch(d["white"], d["black"], d["turn"])
    reach = np.arange(len(status)) < 2586 if len(status) > 2586 else np.ones(len(status), bool)
    won = d["winner"] != 0
    assert np.array_equal(status[won & reach], d["winner"][won & reach])
    nomoves = (d["k"] == 0) & ~won
    assert np.all(status[nomoves] == 2)
    assert np.array_equal(np.packbits(mask.astype(bool), axis=1), d["mask"])


def _check_children(st, d, prefix, idx):
    k = int(d[f"{prefix}_k"][idx])
    assert st is not None
    assert len(st["actions"]) == k
    assert st["root_N"] == int(d[f"{prefix}_root_N"][idx])
    assert np.array_equal(st["actions"], d[f"{prefix}_actions"][idx][:k])
    assert np.array_equal(st["N"], d[f"{prefix}_N"][idx][:k])
    assert np.array_equal(st["W"].view(np.uint32), d[f"{prefix}_W"][idx][:k].view(np.uint32))
    assert np.array_equal(st["p_is64"], d[f"{prefix}_p_is64"][idx][:k])
    assert np.array_equal(st["P"].view(np.uint64), d[f"{prefix}_P"][idx][:k].view(np.uint64))


@pytest.mark.parametrize("S", [1, 2, 50, 100, 800])
def test_tree_fresh_search_bit_exact(tree_tier, S):
    d = tree_tier
    nG = len(d["white"])
    ev = TableEvaluator()
    m = orc.OracleMCTS(nG, c_puct=1.5)
    for g in range(nG):
        m.set_game(g, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
    m.search(ev, S)
    for g in range(nG):
        _check_children(m.root_children(g), d, f"fresh{S}", g)


@pytest.mark.parametrize("tag", ["reuse", "noisy"])
def test_tree_reuse_sequence_bit_exact(tree_tier, tag):
    d = tree_tier
    games = d[f"{tag}_games"]
    S = int(d[f"{tag}_sims"])
    actions = d[f"{tag}_played"]
    nG = len(games)
    ev = TableEvaluator()
    m = orc.OracleMCTS(nG, c_puct=1.5)
    for j, g in enumerate(games):
        m.set_game(j, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
    alive = np.ones(nG, bool)
    saw64 = False
    for ply in range(actions.shape[0]):
        if not alive.any():
            break
        m.set_active(alive)
        noise = d[f"{tag}_noise"][ply] if tag == "noisy" else None
        m.search(ev, S, noise=noise)
        for j in range(nG):
            if not alive[j]:
                continue
            st = m.root_children(j)
            _check_children(st, d, tag, (ply, j))
            saw64 |= bool(st["p_is64"].any())
            a = int(actions[ply, j])
            m.advance(j, a)
            if orc.is_terminal(m.get_game(j)):
                alive[j] = False
    if tag == "noisy":
        assert saw64


@pytest.mark.parametrize("fixture", ["selfplay_tier.npz", "selfplay_wide_tier.npz"])
def test_selfplay_stream_bit_exact(golden_dir, fixture):
    """execute_parallel_episodes (parallel_trainer.py:137-223) replayed with the
    recorded Dirichlet draws and sampled actions."""
    d = np.load(os.path.join(golden_dir, fixture))
    from selfplay_replay import replay_selfplay_oracle
    states, policies, wls = replay_selfplay_oracle(
        int(d["n_games"]), int(d["sims"]), d["dirichlet"], d["choices"], TableEvaluator())
    n = int(d["n_examples"])
    ref_states = np.unpackbits(d["states"])[: n * 192].reshape(n, 3, 8, 8).astype(np.float32)
    assert len(states) == n
    assert np.array_equal(np.array(states), ref_states)
    assert np.array_equal(np.array(policies).view(np.uint32), d["policies"].view(np.uint32))
    assert np.array_equal(np.array(wls), d["wls"])


@pytest.mark.parametrize("tag", ["plain", "noise"])
def test_tree_wide_positions_bit_exact(golden_dir, tag):
    """64 roots from every stage of the game (tree_wide_tier.npz): one search each, without noise
    and with the reference's recorded Dirichlet draws."""
    d = np.load(os.path.join(golden_dir, "tree_wide_tier.npz"))
    nG = len(d["white"])
    m = orc.OracleMCTS(nG, c_puct=1.5)
    for g in range(nG):
        m.set_game(g, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
    noise = d["noise_draws"] if tag == "noise" else None
    m.search(TableEvaluator(), int(d[f"{tag}_sims"]), noise=noise)
    for g in range(nG):
        _check_children(m.root_children(g), d, tag, g)


@pytest.mark.parametrize("tag", ["c07", "c21", "c40", "c21n"])
def test_tree_other_exploration_constants_bit_exact(golden_dir, tag):
    """Exploration constants 0.7, 2.1 and 4.0 (tree_cpuct_tier.npz): 0.7 and 2.1 have no exact
    float32 form, so this pins that the constant enters the score rounded to float32
    (src/mcts.py:304 multiplies a Python float into a float32 prior)."""
    d = np.load(os.path.join(golden_dir, "tree_cpuct_tier.npz"))
    nG = len(d["white"])
    m = orc.OracleMCTS(nG, c_puct=float(d[f"{tag}_c_puct"]))
    for g in range(nG):
        m.set_game(g, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
    noise = d[f"{tag}_draws"] if f"{tag}_draws" in d else None
    m.search(TableEvaluator(), int(d[f"{tag}_sims"]), noise=noise)
    for g in range(nG):
        _check_children(m.root_children(g), d, tag, g)


def replay_deep_tier(m, d):
    """Drives a backend (oracle or device adapter) through tree_deep_tier.npz: 800 simulations per
    move with the recorded root noise, tree reuse over up to 30 plies; every root must equal the
    reference's.  Returns the number of roots compared and the largest root visit count seen."""
    nG = len(d["white"])
    for g in range(nG):
        m.set_game(g, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
    ev = TableEvaluator()
    S = int(d["sims"])
    alive = np.ones(nG, bool)
    checked = top = 0
    for ply in range(d["played"].shape[0]):
        if not alive.any():
            break
        m.set_active(alive.astype(np.uint8))
        m.search(ev, S, noise=d["noise"][ply])
        for g in range(nG):
            if not alive[g]:
                continue
            st = m.root_children(g)
            k = int(d["k"][ply, g])
            assert st is not None and len(st["actions"]) == k and st["root_N"] == int(d["root_N"][ply, g]), (ply, g)
            assert np.array_equal(st["actions"], d["actions"][ply, g][:k])
            assert np.array_equal(st["N"], d["N"][ply, g][:k]), (ply, g)
            assert np.array_equal(st["W"].view(np.uint32), d["W"][ply, g][:k].view(np.uint32)), (ply, g)
            assert np.array_equal(st["p_is64"], d["p_is64"][ply, g][:k])
            assert np.array_equal(st["P"].view(np.uint64), d["P"][ply, g][:k].view(np.uint64)), (ply, g)
            checked += 1
            top = max(top, st["root_N"])
            m.advance(g, int(d["played"][ply, g]))
            if orc.is_terminal(m.get_game(g)):
                alive[g] = False
    return checked, top


def test_tree_deep_reuse_bit_exact(golden_dir):
    d = np.load(os.path.join(golden_dir, "tree_deep_tier.npz"))
    checked, top = replay_deep_tier(orc.OracleMCTS(len(d["white"]), c_puct=1.5), d)
    assert checked == int((d["played"] != 255).sum()) and top > 1500


# ---------------------------------------------------------------- arena / web latency search
def best_by_visits(actions, visits):
    """src/arena.py:257-262, src/web.py:455-460: first child in insertion order with the
    strictly largest visit count."""
    best_action, best_visits = -1, -1
    for a, n in zip(actions, visits):
        if int(n) > best_visits:
            best_visits, best_action = int(n), int(a)
    return best_action


def test_latency_search_matches_reference(golden_dir):
    """MCTS.search(game, add_noise=False) on a fresh root at 200 simulations (src/mcts.py:76-92
    as arena.py and web.py call it), the move they pick from it and the value web.py shows."""
    d = np.load(os.path.join(golden_dir, "latency_tier.npz"))
    ev, S = TableEvaluator(), int(d["sims"])
    for g in range(len(d["white"])):
        m = orc.OracleMCTS(1)
        m.set_game(0, orc.make_pos(d["white"][g], d["black"][g], d["turn"][g]))
        m.search(ev, S)
        st = m.root_children(0)
        _check_children(st, d, "fresh", g)
        assert best_by_visits(st["actions"], st["N"]) == int(d["best_action"][g])
        assert float(np.float32(st["root_W"]) / np.float32(st["root_N"])) == float(d["root_value"][g])


def test_arena_game_matches_reference(golden_dir):
    """A whole Arena.play_game (src/arena.py:231-270): fresh root every move, most visited child."""
    d = np.load(os.path.join(golden_dir, "latency_tier.npz"))
    ev, S = TableEvaluator(), int(d["sims"])
    pos = orc.start_position()
    for ply, want in enumerate(d["game_actions"]):
        assert not orc.is_terminal(pos)
        m = orc.OracleMCTS(1)
        m.set_game(0, pos)
        m.search(ev, S)
        st = m.root_children(0)
        a = best_by_visits(st["actions"], st["N"])
        assert a == int(want), ply
        assert int(st["N"][list(st["actions"]).index(a)]) == int(d["game_best_visits"][ply])
        pos = orc.step_action(pos, a)
    assert orc.is_terminal(pos) and orc.get_result(pos) == tuple(float(x) for x in d["game_result"])
    assert (pos.white, pos.black) == tuple(int(x) for x in d["game_final"])


# ---------------------------------------------------------------- second movegen oracle
def legal3_to_moves(legal3_row, turn):
    """Per-direction canonical source masks -> {(from, to)} on the real board."""
    out = set()
    for d, step in enumerate((8, 7, 9)):
        bits = int(legal3_row[d])
        for sq in range(64):
            if (bits >> sq) & 1:
                to = sq + step
                out.add((sq, to) if turn > 0 else (63 - sq, 63 - to))
    return out


def check_against_baseline_engine(d, legal3, status):
    """legal3 / status of a move generator against src/baseline/state.py's answers."""
    for i in range(len(d["white"])):
        want = {(int(f), int(t)) for f, t in zip(d["frm"][i][: d["k"][i]], d["to"][i][: d["k"][i]])}
        assert legal3_to_moves(legal3[i], int(d["turn"][i])) == want, i
    w, b = d["white"].astype(np.uint64), d["black"].astype(np.uint64)
    # BitboardState.is_terminal (state.py:276-288): a piece on its goal row or a side extinct
    assert np.array_equal((status == 1) | (status == -1), d["terminal"].astype(bool))
    # check_win (state.py:290-311) where only one side has a claim (arbitrary boards may have both)
    white_claim = ((w & np.uint64(0xFF << 56)) != 0) | (b == 0)
    black_claim = ((b & np.uint64(0xFF)) != 0) | (w == 0)
    one = white_claim ^ black_claim
    assert one.sum() > 50
    assert np.array_equal(status[one], d["win"][one])


def test_oracle_movegen_matches_the_baseline_engine(golden_dir):
    d = np.load(os.path.join(golden_dir, "baseline_movegen_tier.npz"))
    legal3, _, _, status = orc.movegen_batch(d["white"], d["black"], d["turn"], want_mask=False, want_planes=False)
    check_against_baseline_engine(d, legal3, status)


def test_leaf_parallel_model_with_one_descent_per_batch_is_the_exact_search():
    """tests/leafpar_model.py (the CPU statement of the opt-in leaf-parallel mode the GPU test
    compares the device with) degenerates to the reference's search when a batch holds one
    descent: no count is ever in flight during a selection.  Root statistics must then equal the
    oracle's - which the fixtures above pin to the reference - bit for bit, with plain and noisy
    roots, including positions whose table rows force the uniform-prior fallback."""
    import leafpar_model as lpm
    from fake_net import TableEvaluator
    ev = TableEvaluator()

    def evaluate(ps):
        planes = np.stack([orc.encode_state(p) for p in ps])
        return ev(planes)

    w, b, t, win = orc.random_positions(3000, 77)
    rs = np.random.RandomState(3)
    checked = 0
    for case in range(12):
        while True:
            i = int(rs.randint(0, 3000))
            p = orc.make_pos(w[i], b[i], t[i], win[i])
            if not orc.is_terminal(p):
                break
        S = int(rs.choice([1, 7, 60, 150]))
        noisy = bool(case % 2)
        m = orc.OracleMCTS(1, c_puct=1.5)
        m.set_game(0, p)
        k = len(orc.legal_actions(p))
        noise = None
        if noisy:
            noise = np.zeros((1, 48), np.float64)
            noise[0, :k] = rs.dirichlet([0.35] * k)
        m.search(ev, S, noise=noise)
        want = m.root_children(0)
        root = lpm.MNode()
        pr, _ = evaluate([p])
        lpm.expand(root, p, pr[0])
        if noisy:
            m0 = orc.OracleMCTS(1, c_puct=1.5)          # the mixed float64 priors as the oracle holds them
            m0.set_game(0, p)
            m0.search(ev, 0, noise=noise)
            for c, p64 in zip(root.children, m0.root_children(0)["P"]):
                c.P64 = float(p64)
        lpm.search(root, p, evaluate, S, 1, c_puct=1.5, noisy_root=noisy)
        assert root.N == want["root_N"] == S
        assert root.actions == list(want["actions"])
        assert [c.N for c in root.children] == list(want["N"])
        assert [np.float32(c.W).view(np.uint32) for c in root.children] == list(want["W"].view(np.uint32))
        checked += 1
    assert checked == 12
