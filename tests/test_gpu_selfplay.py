"""GPU: the fused self-play loop (search with the HIP network, device move
choice, tree reuse, sample queue)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_model():
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(5)
    return AlphaZeroNet(num_blocks=2, num_filters=128).eval()


def test_fused_search_equals_stepwise_search(small_model):
    """az_engine_search (network inside the library) must build exactly the tree
    that the step API builds when fed the same network's outputs."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import HipInference
    nG, S = 32, 64
    inf = HipInference(small_model, max_batch=64)
    w, b, t, win = orc.random_positions(600, seed=9)
    idx = [i for i in range(600) if win[i] == 0][:: 600 // nG][:nG]
    fused, step = Engine(nG, pool_nodes_per_game=1 << 15), Engine(nG, pool_nodes_per_game=1 << 15)
    for e in (fused, step):
        e.set_positions(np.arange(nG), w[idx], b[idx], t[idx])
    fused.attach_net(inf)
    fused.search(S, add_noise=False)

    def evaluate(planes, need):
        probs, value, _, _ = inf.forward_planes(planes)
        return probs, value
    step.search_with(evaluate, S)
    for g in range(nG):
        a, o = fused.root_children(g), step.root_children(g)
        assert np.array_equal(a["actions"], o["actions"])
        assert np.array_equal(a["N"], o["N"])
        assert np.array_equal(a["W"].view(np.uint32), o["W"].view(np.uint32))
        assert a["root_N"] == S


def test_selfplay_games_are_legal_and_samples_consistent(small_model):
    from alphazero_amd.selfplay import SelfPlay
    nG, S = 16, 24
    sp = SelfPlay(small_model, n_games=nG, num_simulations=S, seed=3, pool_nodes_per_game=1 << 15)
    e = sp.engine
    e.reset_all()
    pos = [orc.start_position() for _ in range(nG)]
    plies = np.zeros(nG, int)
    finished = 0
    states, policies, wls = [], [], []
    for it in range(400):
        w0, b0, t0, _ = e.get_positions()
        e.play_move(S, add_noise=True, restart_finished=False)
        acts = e.chosen_actions()
        w1, b1, t1, win1 = e.get_positions()
        for g in range(nG):
            if acts[g] < 0:
                continue
            assert acts[g] in orc.legal_actions(pos[g]), "engine played an illegal move"
            pos[g] = orc.step_action(pos[g], int(acts[g]))
            plies[g] += 1
            assert (pos[g].white, pos[g].black, pos[g].turn, pos[g].winner) == (int(w1[g]), int(b1[g]), int(t1[g]), int(win1[g]))
        st = e.stats()
        if st["games_finished"] == nG:
            break
    assert e.stats()["games_finished"] == nG
    s, p, wl = [x.cpu().numpy() for x in e.drain_samples()]
    assert len(s) == plies.sum() == e.stats()["moves"]
    assert np.allclose(p.sum(1), 1.0, atol=1e-5)
    assert set(np.unique(s)) <= {0.0, 1.0} and np.all(s[:, 2] == 1.0)
    assert np.all(wl.sum(1) == 1.0)
    # every game is decisive: per game the winners' and losers' plies alternate
    for g in range(nG):
        assert orc.is_terminal(pos[g])
    # after ply 16 the policy is one-hot (temperature 0)
    onehot = (p.max(1) == 1.0).sum()
    assert onehot >= (plies - 16).clip(0).sum()


def test_continuous_selfplay_restarts_and_drains(small_model):
    from alphazero_amd.selfplay import SelfPlay
    sp = SelfPlay(small_model, n_games=8, num_simulations=8, seed=1, pool_nodes_per_game=1 << 14)
    for _ in range(150):
        sp.step(restart_finished=True)
    st = sp.engine.stats()
    assert st["moves"] == 150 * 8
    assert st["games_finished"] >= 8
    s, p, wl = sp.engine.drain_samples()
    assert len(s) == st["samples_pending"] > 0
    assert sp.engine.stats()["samples_pending"] == 0
    assert st["simulations"] == 150 * 8 * 8


def test_full_size_plies_keep_the_search_invariants():
    """The benchmark configuration itself (1024 games x 800 simulations, 20 x 128 network, root
    noise): size-independent properties of three plies with tree reuse, checked for every game."""
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import SelfPlay
    nG, S = 1024, 800
    torch.manual_seed(0)
    sp = SelfPlay(AlphaZeroNet(num_blocks=20, num_filters=128).eval(), n_games=nG, num_simulations=S, seed=77)
    e = sp.engine
    pos = [orc.start_position() for _ in range(nG)]
    carried = np.zeros(nG, np.int64)              # visits the kept subtree brought into this ply
    slots = np.arange(nG)
    for ply in range(3):
        e.search(S, add_noise=True)
        roots = e.roots_children(slots)
        for g, q in enumerate(roots):
            legal = orc.legal_actions(pos[g])
            assert np.array_equal(q["actions"], legal)                   # children = legal moves, insertion order
            # a fresh root's value is not backed up; a reused root keeps the visits it had as a child
            assert q["root_N"] == carried[g] + S
            # a reused child was visited once as a leaf before it had children of its own
            assert int(q["N"].sum()) == q["root_N"] - (1 if carried[g] > 0 else 0)
            assert np.all(q["p_is64"] == 1) and abs(q["P"].sum() - 1.0) < 1e-6   # noise keeps the priors normalised
        e.play_move(0, add_noise=False, restart_finished=False)           # record, choose, step, re-root
        acts = e.chosen_actions()
        w1, b1, t1, win1 = e.get_positions()
        for g, q in enumerate(roots):
            a = int(acts[g])
            i = int(np.flatnonzero(q["actions"] == a)[0])
            assert q["N"][i] > 0                                          # temperature 1: only visited moves
            carried[g] = int(q["N"][i])
            pos[g] = orc.step_action(pos[g], a)
            assert (pos[g].white, pos[g].black, pos[g].turn) == (int(w1[g]), int(b1[g]), int(t1[g]))
            assert pos[g].winner == 0 and int(win1[g]) == 0               # nobody wins in three plies
    assert e.pool_usage().max() < (1 << 18)
    st = e.stats()
    assert st["moves"] == 3 * nG and st["simulations"] == 3 * nG * S
    s, p, wl = e.drain_samples()
    assert len(s) == 0                                                    # no game has finished


def test_full_size_fused_search_equals_the_oracle_on_sampled_games():
    """The benchmark's own size and kernel form: 1024 games x 800 simulations through the fused
    search with the 20 x 128 network (the large-batch tower at 1024 rows, device row assignment)
    and injected root noise, two plies with tree reuse.  Eight games spread over the batch are
    replayed by the CPU oracle, which is given the same network's outputs through the same
    tower form at 8 rows (a position's outputs do not depend on its row or its neighbours):
    identical visit counts, value sums and priors.  Reference: MCTS.search_batch
    (src/mcts.py:94-189)."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet, HipInference
    nG, S = 1024, 800
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=20, num_filters=128).eval()
    inf = HipInference(model, max_batch=nG)
    w, b, t, win = orc.random_positions(4000, seed=33)
    idx = [i for i in range(4000) if win[i] == 0 and orc.legal_actions(orc.make_pos(w[i], b[i], t[i])).size][:nG]
    assert len(idx) == nG
    idx = np.asarray(idx)
    dev = Engine(nG)
    dev.set_positions(np.arange(nG), w[idx], b[idx], t[idx])
    dev.attach_net(inf)
    sel = np.array([0, 1, 127, 300, 511, 512, 777, 1023])
    cpu = orc.OracleMCTS(len(sel))
    for j, g in enumerate(sel):
        cpu.set_game(j, orc.make_pos(w[idx[g]], b[idx[g]], t[idx[g]]))
    small = HipInference(model, max_batch=16)
    small.set_tower_form(16)                      # the form the 1024-row launches use
    start = orc.encode_state(orc.start_position())

    def evaluate(planes, need):
        x = np.ascontiguousarray(planes, np.float32).copy()
        x[np.asarray(need) == 0] = start
        probs, value, _, _ = small.forward_planes(torch.from_numpy(x).cuda())
        return probs.cpu().numpy().copy(), value.cpu().numpy().copy()

    rs = np.random.RandomState(11)
    pos = [orc.make_pos(w[i], b[i], t[i]) for i in idx]
    alive = np.ones(nG, bool)
    compared = 0
    for ply in range(2):
        dev.set_active(alive.astype(np.uint8)); cpu.set_active(alive[sel])
        noise = rs.dirichlet([0.35] * 48, size=nG)
        dev.search(0, add_noise=False)                # expands the roots that need it
        dev.add_noise(noise)
        dev.search(S, add_noise=False)
        cpu.search(evaluate, S, noise=noise[sel])
        for j, g in enumerate(sel):
            if not alive[g]:
                continue
            a, o = dev.root_children(int(g)), cpu.root_children(j)
            assert np.array_equal(a["actions"], o["actions"])
            assert np.array_equal(a["N"], o["N"]), (ply, int(g))
            assert np.array_equal(a["W"].view(np.uint32), o["W"].view(np.uint32)), (ply, int(g))
            assert np.array_equal(a["P"].view(np.uint64), o["P"].view(np.uint64)), (ply, int(g))
            compared += 1
        # every live game plays its most-visited move (first in insertion order among ties)
        acts = np.full(nG, -1, np.int16)
        live = np.flatnonzero(alive)
        for g, q in zip(live, dev.roots_children(live)):
            acts[g] = int(q["actions"][np.argmax(q["N"])])
            pos[g] = orc.step_action(pos[g], int(acts[g]))
        for j, g in enumerate(sel):
            if alive[g]:
                cpu.advance(j, int(acts[g]))
        dev.advance_all(acts)
        for g in live:
            if orc.is_terminal(pos[g]):
                alive[g] = False
    assert compared >= 12
    small.close()


@pytest.mark.parametrize("net", ["2x128", "2x32", "1x64 se 4"])
def test_fused_search_with_the_real_network_equals_the_oracle(small_model, net):
    """The whole fused path (HIP network inside the library, 800 simulations, injected root
    noise, tree reuse over two plies) against the CPU oracle driven with the same network's
    outputs: identical visit counts, value sums and priors for every root.  Also with the
    reference's test network (2 x 32, tests/test_mcts.py:54) and another width, which run the
    any-width kernels with the device row assignment."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet, HipInference
    nG, S = 8, 800
    torch.manual_seed(8)
    model = {"2x128": small_model, "2x32": AlphaZeroNet(num_blocks=2, num_filters=32).eval(),
             "1x64 se 4": AlphaZeroNet(num_blocks=1, num_filters=64, se_ratio=4).eval()}[net]
    inf = HipInference(model, max_batch=16)
    w, b, t, win = orc.random_positions(900, seed=21)
    idx = [i for i in range(900) if win[i] == 0 and orc.legal_actions(orc.make_pos(w[i], b[i], t[i])).size][:: 900 // nG][:nG]
    dev = Engine(nG, pool_nodes_per_game=1 << 16)
    dev.set_positions(np.arange(nG), w[idx], b[idx], t[idx])
    dev.attach_net(inf)
    cpu = orc.OracleMCTS(nG)
    for j, i in enumerate(idx):
        cpu.set_game(j, orc.make_pos(w[i], b[i], t[i]))
    start = orc.encode_state(orc.start_position())

    def evaluate(planes, need):
        x = np.ascontiguousarray(planes, np.float32).copy()
        x[np.asarray(need) == 0] = start                  # rows the oracle does not ask for: any valid state
        probs, value, _, _ = inf.forward_planes(torch.from_numpy(x).cuda())
        return probs.cpu().numpy().copy(), value.cpu().numpy().copy()

    rs = np.random.RandomState(4)
    alive = np.ones(nG, bool)
    for ply in range(2):
        dev.set_active(alive.astype(np.uint8)); cpu.set_active(alive)
        noise = rs.dirichlet([0.35] * 48, size=nG)
        dev.search(0, add_noise=False)                    # expands the roots that need it
        dev.add_noise(noise)
        dev.search(S, add_noise=False)
        cpu.search(evaluate, S, noise=noise)
        for g in range(nG):
            if not alive[g]:
                continue
            a, o = dev.root_children(g), cpu.root_children(g)
            assert np.array_equal(a["actions"], o["actions"])
            assert np.array_equal(a["N"], o["N"]), (ply, g)
            assert np.array_equal(a["W"].view(np.uint32), o["W"].view(np.uint32))
            assert np.array_equal(a["P"].view(np.uint64), o["P"].view(np.uint64))
        acts = np.full(nG, -1, np.int16)
        for g in range(nG):
            if alive[g]:
                o = cpu.root_children(g)
                acts[g] = int(o["actions"][np.argmax(o["N"])])
                cpu.advance(g, int(acts[g]))
                if orc.is_terminal(cpu.get_game(g)):
                    alive[g] = False
        dev.advance_all(acts)


def test_selfplay_is_reproducible_for_a_seed(small_model):
    """Two engines with the same seed play the same games (device RNG is counter-based);
    another seed plays different ones."""
    from alphazero_amd.selfplay import SelfPlay

    def run(seed):
        sp = SelfPlay(small_model, n_games=32, num_simulations=32, seed=seed, pool_nodes_per_game=1 << 14)
        moves = []
        for _ in range(12):
            sp.step(restart_finished=True)
            moves.append(sp.engine.chosen_actions().copy())
        return np.stack(moves), [x.cpu().numpy() for x in sp.engine.drain_samples()]
    a_moves, a_samples = run(9)
    b_moves, b_samples = run(9)
    c_moves, _ = run(10)
    assert np.array_equal(a_moves, b_moves)
    for x, y in zip(a_samples, b_samples):
        assert np.array_equal(x, y)
    assert not np.array_equal(a_moves, c_moves)


def test_fused_search_packs_the_active_games(small_model):
    """With some games inactive the fused search evaluates only the active ones, packed into a
    smaller network batch; the trees must equal those of the step API, which evaluates one
    row per game, and inactive games must stay untouched."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import HipInference
    nG, S = 32, 64
    inf = HipInference(small_model, max_batch=64)
    w, b, t, win = orc.random_positions(600, seed=9)
    idx = [i for i in range(600) if win[i] == 0][:: 600 // nG][:nG]
    active = np.zeros(nG, np.uint8)
    active[[0, 3, 4, 9, 17, 18, 19, 30, 31]] = 1
    fused, step = Engine(nG, pool_nodes_per_game=1 << 15), Engine(nG, pool_nodes_per_game=1 << 15)
    for e in (fused, step):
        e.set_positions(np.arange(nG), w[idx], b[idx], t[idx])
        e.set_active(active)
    fused.attach_net(inf)
    fused.search(S, add_noise=False)

    start = torch.from_numpy(orc.encode_state(orc.start_position())).cuda()

    def evaluate(planes, need):
        x = planes.clone()
        x[need == 0] = start             # rows nobody asks for (inactive games hold no planes): any valid state
        probs, value, _, _ = inf.forward_planes(x)
        return probs, value
    step.search_with(evaluate, S)
    for g in range(nG):
        a, o = fused.root_children(g), step.root_children(g)
        if not active[g]:
            assert a is None and o is None
            continue
        assert np.array_equal(a["actions"], o["actions"])
        assert np.array_equal(a["N"], o["N"])
        assert np.array_equal(a["W"].view(np.uint32), o["W"].view(np.uint32))
        assert np.array_equal(a["P"].view(np.uint64), o["P"].view(np.uint64))
        assert a["root_N"] == S
    active[:] = 0
    fused.set_active(active)
    fused.search(S)                      # nothing active: a no-op, not an error


@pytest.mark.parametrize("fixture", ["selfplay_tier.npz", "selfplay_wide_tier.npz"])
def test_fused_sample_stream_equals_the_reference_stream(golden_dir, small_model, fixture):
    """The product kernels of the self-play bookkeeping (record-and-choose, finish scan, sample
    queue, unpack on drain: az_engine_play_move + az_engine_drain_samples) against the stream the
    reference's execute_parallel_episodes produced with a table network and recorded numpy draws
    (tests/golden/make_golden.py selfplay): every (state, pi, wl) example, in the reference's
    order.  The search of each ply runs through the step API with the table network and the
    recorded Dirichlet rows; play_move then does everything after the search with the recorded
    moves injected."""
    import os
    from fake_net import TableEvaluator
    from gpu_backend import HostEvaluatorOnDevice
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import HipInference
    d = np.load(os.path.join(golden_dir, fixture))
    nG, S = int(d["n_games"]), int(d["sims"])
    e = Engine(nG, pool_nodes_per_game=1 << 14)
    # play_move wants a network attached; every root is already expanded when it is called
    # here, so nothing the network computes is consumed
    e.attach_net(HipInference(small_model, max_batch=max(16, nG)))
    e.reset_all()
    evaluate = HostEvaluatorOnDevice(TableEvaluator(), e.device)
    pos = [orc.start_position() for _ in range(nG)]
    active = list(range(nG))
    di = ci = 0
    while active:
        mask = np.zeros(nG, np.uint8)
        mask[active] = 1
        e.set_active(mask)
        noise = np.zeros((nG, 48))
        acts = np.full(nG, -1, np.int16)
        for i in active:                       # one Dirichlet draw and one move per active game, in order
            noise[i] = d["dirichlet"][di]
            di += 1
        e.search_with(evaluate, S, noise=noise)
        for i in active:
            acts[i] = d["choices"][ci]
            ci += 1
        e.play_move(0, add_noise=False, actions=acts, restart_finished=False)
        for i in active:
            pos[i] = orc.step_action(pos[i], int(acts[i]))
        active = [i for i in active if not orc.is_terminal(pos[i])]
    assert di == len(d["dirichlet"]) and ci == len(d["choices"])
    st = e.stats()
    assert st["games_finished"] == nG and st["moves"] == len(d["choices"])
    n = int(d["n_examples"])
    assert e.samples_pending() == n
    s, p, wl = [x.cpu().numpy() for x in e.drain_samples()]
    want_s = np.unpackbits(d["states"])[: n * 192].reshape(n, 3, 8, 8).astype(np.float32)
    assert np.array_equal(s, want_s)
    assert np.array_equal(p.view(np.uint32), d["policies"].view(np.uint32))
    assert np.array_equal(wl, d["wls"])


def test_bench_starts_its_own_ranks_on_the_gpu():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the script starts two ranks itself, both
    play on this box's one GPU (gloo rehearsal of the collectives) and rank 0's single JSON
    line is relayed with the whole-job counts.  The CPU baseline leg runs once per job and is
    part of that line: in the parent when bench.py starts the ranks, on rank 0 (before it joins
    the process group) when a launcher started them, as the round-end driver does."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["AZ_BENCH_BACKEND"] = "gloo"
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--games", "32", "--sims", "16", "--blocks", "2",
            "--cpu-seconds", "1.5"]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                "127.0.0.1", "--master-port", "29731"]
    for cmd in ([sys.executable, os.path.join(root, "bench.py")] + args, launcher + [os.path.join(root, "bench.py")] + args):
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip().startswith('{"metric"')]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
        # two ranks x 32 games x 16 simulations x 2 plies
        assert round(out["value"] * out["ms_per_step"] * 1e-3 * 2) == 2 * 32 * 16 * 2
        assert "error" not in out["exchange"] and out["exchange"]["backend"] == "gloo"
        cpu = out["cpu_baseline"]
        assert cpu["kind"] == "port" and cpu["unit"] == "sims/s" and cpu["value"] > 0 and cpu["cores"] >= 1


def test_packed_samples_unpack_to_the_drained_arrays(small_model):
    """Engine.drain_packed + unpack_samples (the wire format of the exchange between GPUs) give
    exactly what az_engine_drain_samples gives: two engines with one seed play the same games."""
    from alphazero_amd import _lib
    from alphazero_amd.engine import unpack_samples
    from alphazero_amd.selfplay import SelfPlay

    def play():
        sp = SelfPlay(small_model, n_games=24, num_simulations=12, seed=21, pool_nodes_per_game=1 << 14)
        for _ in range(90):
            sp.step(restart_finished=True)
        return sp.engine
    a, b = play(), play()
    n = a.samples_pending()
    assert n == b.samples_pending() and n > 200
    want = [t.cpu().numpy() for t in a.drain_samples()]
    packed = b.drain_packed()
    assert packed.shape == (n, _lib.PACKED_SAMPLE_BYTES) and b.samples_pending() == 0
    got = [t.cpu().numpy() for t in unpack_samples(packed)]
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    # rows are self-contained: any subset, in any order, unpacks to the same rows
    pick = torch.tensor([n - 1, 0, n // 2], device=packed.device)
    sub = [t.cpu().numpy() for t in unpack_samples(packed[pick].contiguous())]
    for g, w in zip(sub, want):
        assert np.array_equal(g, w[pick.cpu().numpy()])
    assert unpack_samples(packed[:0])[0].shape == (0, 3, 8, 8)


def test_network_rows_are_exactly_the_evaluated_leaves(small_model):
    """The network launches of the fused search carry one row per leaf that needs an evaluation:
    idle games and terminal leaves (backed up without the network, src/mcts.py:183-187) take
    none.  Late in a game most simulations end in terminal leaves."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import HipInference
    nG, S = 37, 200                                   # not a multiple of 16 or of 4
    w, b, t, win = orc.random_positions(4000, seed=5)
    # a piece one step from its goal row: wins are one or two plies below the root
    late = [i for i in range(4000) if win[i] == 0 and ((int(w[i]) >> 48) & 0xFF or (int(b[i]) >> 8) & 0xFF)
            and orc.legal_actions(orc.make_pos(w[i], b[i], t[i])).size][:nG]
    assert len(late) == nG
    e = Engine(nG, pool_nodes_per_game=1 << 15)
    e.attach_net(HipInference(small_model, max_batch=64))
    e.set_positions(np.arange(nG), w[late], b[late], t[late])
    active = np.ones(nG, np.uint8)
    active[[3, 11, 36]] = 0
    e.set_active(active)
    e.search(S, add_noise=True)
    st = e.stats()
    assert st["simulations"] == 34 * S
    assert st["network_rows"] == st["nn_evals"]
    assert st["nn_evals"] < 0.9 * (st["simulations"] + 34)        # many leaves were terminal
    assert st["nn_evals"] >= 34                                     # every root was evaluated once


def test_fused_search_with_more_games_than_one_scan_pass(small_model):
    """Row assignment scans 16 flags per lane and pass (4096 games per pass for the 256-lane tower
    forms): an engine with more games than that must give the trees of the step API too."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import HipInference
    nG, S = 4100 + 64, 3
    inf = HipInference(small_model, max_batch=nG)
    w, b, t, win = orc.random_positions(nG + 600, seed=17)
    idx = [i for i in range(len(w)) if win[i] == 0][:nG]
    rs = np.random.RandomState(2)
    active = (rs.rand(nG) < 0.8).astype(np.uint8)
    active[-1] = active[0] = active[4095] = active[4096] = 1
    fused, step = Engine(nG, pool_nodes_per_game=256), Engine(nG, pool_nodes_per_game=256)
    for e in (fused, step):
        e.set_positions(np.arange(nG), w[idx], b[idx], t[idx])
        e.set_active(active)
    fused.attach_net(inf)
    fused.search(S, add_noise=False)
    start = torch.from_numpy(orc.encode_state(orc.start_position())).cuda()

    def evaluate(planes, need):
        x = planes.clone()
        x[need == 0] = start
        probs, value, _, _ = inf.forward_planes(x)
        return probs, value
    step.search_with(evaluate, S)
    slots = np.flatnonzero(active)
    for a, o in zip(fused.roots_children(slots), step.roots_children(slots)):
        assert np.array_equal(a["actions"], o["actions"]) and np.array_equal(a["N"], o["N"])
        assert np.array_equal(a["W"].view(np.uint32), o["W"].view(np.uint32))
    st = fused.stats()
    assert st["network_rows"] == st["nn_evals"]


def test_sample_queue_order_with_more_games_than_one_scan_pass(small_model):
    """1,100 games (the finish scan handles 1,024 per pass) played to the end with device-drawn
    moves: the drained samples must come ply by ply and, within a ply, by ascending game index,
    each finished game contributing its whole history in move order with the right wl sign -
    checked against positions replayed on the host from the moves the engine reports."""
    from alphazero_amd.selfplay import SelfPlay
    nG = 1100
    sp = SelfPlay(small_model, n_games=nG, num_simulations=1, seed=12, pool_nodes_per_game=256)
    e = sp.engine
    e.reset_all()
    e.set_active(np.ones(nG, np.uint8))
    pos = [orc.start_position() for _ in range(nG)]
    hist = [[] for _ in range(nG)]
    alive = np.ones(nG, bool)
    want_states, want_wl = [], []
    for _ in range(300):
        if not alive.any():
            break
        e.play_move(1, add_noise=True, restart_finished=False)
        acts = e.chosen_actions()
        for g in np.flatnonzero(alive):
            assert acts[g] >= 0
            hist[g].append((orc.encode_state(pos[g]), pos[g].turn))
            pos[g] = orc.step_action(pos[g], int(acts[g]))
            if orc.is_terminal(pos[g]):
                w, l = orc.get_result(pos[g])
                for s, turn in hist[g]:
                    want_states.append(s)
                    want_wl.append((w, l) if turn == 1 else (l, w))
                alive[g] = False
    assert not alive.any()
    s, p, wl = [x.cpu().numpy() for x in e.drain_samples()]
    assert len(s) == len(want_states) == e.stats()["moves"]
    assert np.array_equal(s, np.stack(want_states))
    assert np.array_equal(wl, np.array(want_wl, np.float32))
    assert np.allclose(p.sum(1), 1.0, atol=1e-5)


def test_exchange_steps_run_over_rccl_with_one_rank():
    """The exchange code over the real RCCL backend (backend "nccl"), world of one rank: the
    collectives' dtypes and call forms are ones RCCL takes.  Own process: a process group is
    process-wide state."""
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    here = os.path.dirname(os.path.abspath(__file__))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(here, "nccl_single_rank.py"), str(port)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-500:], r.stderr[-3000:])
