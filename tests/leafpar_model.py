"""CPU model of the leaf-parallel latency search (az_engine_search_leaf_parallel, csrc/leafpar.hip).

TEST INFRASTRUCTURE.  The mode has no counterpart in the reference (src/mcts.py:153-187 evaluates
one leaf per game and simulation), so there is no reference output to pin it to; this file states
its semantics a second time, in plain Python with numpy float32 / float64 scalars where the device
code uses them, and the GPU test compares whole trees bit for bit:

  * a batch is up to K descents of one game, made one after the other before the network runs;
  * every node on the path of a pending descent carries one in-flight count; selection reads a
    child as N + in_flight visits and W + in_flight value (a pending descent counts as a loss for
    the player who chose it), everything else is _select_child (src/mcts.py:281-312): float32
    arithmetic, float64 at a noisy root and under a uniform-prior node, ties to the first child;
  * a descent that stops at the leaf of an earlier descent of the same batch shares its value;
  * then, in descent order: expand (src/mcts.py:260-279), back up (src/mcts.py:337-342) and
    take the in-flight counts away again.

Game rules come from the C oracle (legal moves in insertion order, step, terminal test)."""
import math

import numpy as np

from oracle import oracle as orc

F32 = np.float32


class MNode:
    __slots__ = ("N", "W", "P", "P64", "children", "actions", "uniform", "vl")

    def __init__(self, prior=F32(0)):
        self.N, self.W, self.P, self.P64 = 0, F32(0), F32(prior), None
        self.children, self.actions, self.uniform, self.vl = [], [], False, 0


def expand(node, pos, probs):
    """_expand_node: sequential float32 prior sum over the legal moves in insertion order."""
    acts = orc.legal_actions(pos)
    prior_sum = F32(0)
    for a in acts:
        prior_sum = F32(prior_sum + F32(probs[a]))
    node.actions = [int(a) for a in acts]
    if prior_sum > 0:
        node.children = [MNode(F32(F32(probs[a]) / prior_sum)) for a in acts]
    else:                                   # uniform fallback: the children hold the float64 1/k
        node.uniform = True
        node.children = [MNode(F32(1.0 / len(acts))) for _ in acts]


def terminal_value(pos):
    """Seen by the side to move at the leaf (src/mcts.py:328-335)."""
    if pos.winner != 0:
        return F32(1.0) if pos.winner == pos.turn else F32(-1.0)
    return F32(-1.0)


def select_child(node, pn, c_puct, at_noisy_root):
    cp32 = F32(c_puct)
    sqrt_parent = math.sqrt(float(max(pn, 1)))
    k = len(node.children)
    f64mode = at_noisy_root or node.uniform
    best, best_score = 0, -math.inf
    for i, c in enumerate(node.children):
        n_eff = c.N + c.vl
        w_eff = F32(c.W + F32(c.vl))
        if not f64mode:
            q = F32(0) if n_eff == 0 else F32(-F32(w_eff / F32(n_eff)))
            u = F32(cp32 * c.P)
            u = F32(u * F32(sqrt_parent))
            u = F32(u / F32(1 + n_eff))
            score = float(F32(q + u))
        else:
            p64 = float(c.P64) if at_noisy_root else 1.0 / k
            q = 0.0 if n_eff == 0 else float(F32(-F32(w_eff / F32(n_eff))))
            u = c_puct * p64
            u = u * sqrt_parent
            u = u / float(1 + n_eff)
            score = q + u
        if score > best_score:              # strict: the first child wins ties
            best, best_score = i, score
    return best


def search(root, root_pos, evaluate, num_simulations, leaf_batch, c_puct=1.5, noisy_root=False):
    """num_simulations descents from `root` (expanded, position root_pos) in batches of leaf_batch.
    evaluate(list of positions) -> (probs [n][192] float32, values [n] float32).
    Returns how many descents ended at a terminal leaf and how many shared a leaf of their batch."""
    stats = dict(terminal=0, shared=0, evaluated=0)
    done = 0
    while done < num_simulations:
        K = min(leaf_batch, num_simulations - done)
        rows = []
        for j in range(K):
            node, pos, path = root, root_pos, [root]
            pn = node.N + node.vl
            node.vl += 1
            while node.children:
                b = select_child(node, pn, c_puct, noisy_root and node is root)
                child = node.children[b]
                pn = child.N + child.vl
                child.vl += 1
                pos = orc.step_action(pos, node.actions[b])
                node = child
                path.append(node)
            same = next((i for i, r in enumerate(rows) if r["node"] is node), None)
            kind = 2 if same is not None else (1 if orc.is_terminal(pos) else 0)
            rows.append(dict(node=node, pos=pos, path=path, kind=kind, first=same if same is not None else j))
            stats[("evaluated", "terminal", "shared")[kind]] += 1
        todo = [r for r in rows if r["kind"] == 0]
        if todo:
            probs, values = evaluate([r["pos"] for r in todo])
            for r, p, v in zip(todo, probs, values):
                r["probs"], r["value"] = p, F32(v)
        vals = []
        for r in rows:
            if r["kind"] == 0:
                expand(r["node"], r["pos"], r["probs"])
                v = r["value"]
            elif r["kind"] == 1:
                v = terminal_value(r["pos"])
            else:
                v = vals[r["first"]]
            vals.append(v)
            n = len(r["path"])
            for i, nd in enumerate(r["path"]):
                vv = F32(-v) if ((n - 1 - i) & 1) else v
                nd.N += 1
                nd.W = F32(nd.W + vv)
                nd.vl -= 1
        done += K
    return stats
