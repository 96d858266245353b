"""GPU: the leaf-parallel latency search (az_engine_search_leaf_parallel) against its CPU model
(tests/leafpar_model.py), whole trees bit for bit.

The mode has no counterpart in the reference (its results differ from MCTS.search by
construction), so it cannot be pinned to a reference output; the model restates its semantics
independently of the kernels - in-flight counts as virtual losses, shared leaves inside a batch,
expansion and backup in descent order - with the reference's float32 / float64 arithmetic for
everything else (src/mcts.py:260-342).  Both sides evaluate leaves with the same HIP network
(a position's result does not depend on its row or batch: tests/test_gpu_net.py)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

import leafpar_model as lp

pytestmark = pytest.mark.gpu


def _positions(n, seed):
    """n live mid-game positions of seeded random play, both colours to move."""
    w, b, t, win = orc.random_positions(4000, seed)
    rs = np.random.RandomState(seed)
    out = []
    for want_turn in ([1, -1] * n)[:n]:
        while True:
            i = int(rs.randint(200, 4000))
            p = orc.make_pos(w[i], b[i], t[i], win[i])
            if t[i] == want_turn and not orc.is_terminal(p) and bin(int(w[i])).count("1") + bin(int(b[i])).count("1") < 28:
                out.append(p)
                break
    return out


def _compare(engine, g, idx, node, where, noisy_root=False):
    q = engine.node_children(g, idx)
    assert q is not None, where
    assert q["root_N"] == node.N, (where, q["root_N"], node.N)
    assert np.float32(q["root_W"]).view(np.uint32) == np.float32(node.W).view(np.uint32), (where, q["root_W"], node.W)
    assert list(q["actions"]) == node.actions, where
    k = len(node.children)
    for i, c in enumerate(node.children):
        assert int(q["N"][i]) == c.N, (where, i, int(q["N"][i]), c.N)
        assert q["W"][i].view(np.uint32) == np.float32(c.W).view(np.uint32), (where, i, q["W"][i], c.W)
        if noisy_root:
            assert q["p_is64"][i] and q["P"][i] == float(c.P64), (where, i)
        elif node.uniform:
            assert q["p_is64"][i] and q["P"][i] == 1.0 / k, (where, i)
        else:
            assert not q["p_is64"][i] and np.float32(q["P"][i]).view(np.uint32) == np.float32(c.P).view(np.uint32), (where, i)
    count = 1
    for i, c in enumerate(node.children):
        if c.N > 0 or c.children:
            count += _compare(engine, g, int(q["index"][i]), c, f"{where}/{node.actions[i]}")
    return count


def run_case(K, S, noise, seed, blocks=2, scale=6.0, n=4, c_puct=1.5):
    """One comparison: n games, S simulations in batches of K; returns (nodes compared, model statistics)."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet, HipInference
    torch.manual_seed(seed)
    model = AlphaZeroNet(num_blocks=blocks, num_filters=128).eval()
    model.policy_fc.weight.data.mul_(scale)        # peaked priors: the descents of a batch collide on leaves
    inf = HipInference(model, max_batch=256)
    positions = _positions(n, seed=100 + seed)
    engine = Engine(n + 1, c_puct=c_puct, pool_nodes_per_game=1 << 14)       # one slot stays idle
    engine.attach_net(inf)
    slots = np.arange(n, dtype=np.int32)
    engine.set_positions(slots, [p.white for p in positions], [p.black for p in positions], [p.turn for p in positions])
    active = np.zeros(n + 1, np.uint8)
    active[:n] = 1
    engine.set_active(active)
    engine.search(0)                                          # expands the roots with the HIP network

    def evaluate(ps):
        x = torch.from_numpy(np.stack([orc.encode_state(p) for p in ps])).cuda()
        probs, value, _, _ = inf.forward_planes(x)
        return probs.cpu().numpy(), value.cpu().numpy()

    roots = []
    for p in positions:
        r = lp.MNode()
        pr, _ = evaluate([p])
        lp.expand(r, p, pr[0])
        roots.append(r)
    if noise:
        rs = np.random.RandomState(seed)
        nz = np.zeros((n + 1, 48), np.float64)
        for g, r in enumerate(roots):
            nz[g, :len(r.children)] = rs.dirichlet([0.35] * len(r.children))
        engine.add_noise(nz)
        for g, r in enumerate(roots):                         # the mixed float64 priors (pinned by the tree tiers) as the device holds them
            q = engine.root_children(g)
            assert q["p_is64"].all()
            for c, p64 in zip(r.children, q["P"]):
                c.P64 = float(p64)
    engine.search_leaf_parallel(slots, S, K)
    engine.pool_usage()                                       # reports a pending device error
    total_nodes = 0
    stats = dict(terminal=0, shared=0, evaluated=0)
    for g, (r, p) in enumerate(zip(roots, positions)):
        st = lp.search(r, p, evaluate, S, K, c_puct=c_puct, noisy_root=noise)
        for key in stats:
            stats[key] += st[key]
        assert r.N == S and r.vl == 0
        total_nodes += _compare(engine, g, -1, r, f"game {g}", noisy_root=noise)
    assert engine.root_children(n) is None                    # the idle slot was left alone
    engine.close()
    inf.close()
    return total_nodes, stats


@pytest.mark.parametrize("K,S,noise", [(2, 37, False), (5, 64, False), (16, 128, False), (32, 100, False), (8, 64, True),
                                       (32, 33, True)])
def test_leaf_parallel_trees_equal_the_cpu_model(K, S, noise):
    nodes, stats = run_case(K, S, noise, seed=7 + K)
    print(f"K {K} S {S} noise {noise}: {nodes} nodes compared, descents: {stats}")
    assert stats["evaluated"] + stats["terminal"] + stats["shared"] == 4 * S
    if K >= 16:
        assert stats["shared"] > 0          # the case exercises leaves shared inside a batch
