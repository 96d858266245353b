"""Python handle on the device search engine (az_engine_* in include/azamd.h).

PyTorch is used for device memory and the current HIP stream only; every
computation happens in libazamd.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


def _dev_ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _host_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _stream(device=None):
    """The caller's current HIP stream on `device` (default: the current device)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def unpack_samples(packed):
    """Packed sample rows (uint8 cuda tensor [n, 792], from any rank's Engine.drain_packed)
    -> (states[n,3,8,8], policies[n,192], wls[n,2]) float32 cuda tensors."""
    assert packed.is_cuda and packed.dtype == torch.uint8 and packed.is_contiguous()
    assert packed.dim() == 2 and packed.shape[1] == _lib.PACKED_SAMPLE_BYTES
    n, dev = packed.shape[0], packed.device
    states = torch.empty((n, 3, 8, 8), dtype=torch.float32, device=dev)
    policies = torch.empty((n, 192), dtype=torch.float32, device=dev)
    wls = torch.empty((n, 2), dtype=torch.float32, device=dev)
    if n:
        with torch.cuda.device(dev):
            check(_lib.lib().az_unpack_samples(n, _dev_ptr(packed), _dev_ptr(states), _dev_ptr(policies),
                                               _dev_ptr(wls), _stream(dev)))
    return states, policies, wls


class Engine:
    """B games, one tree each, resident on one GPU."""

    def __init__(self, n_games, c_puct=1.5, dirichlet_alpha=0.35, dirichlet_epsilon=0.25,
                 temperature_threshold=16, pool_nodes_per_game=0, seed=0, device=None):
        if not torch.cuda.is_available():
            raise _lib.AzError("alphazero_amd.Engine needs a GPU (no CPU fallback exists)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.n = int(n_games)
        cfg = _lib.EngineConfig(self.n, float(c_puct), float(dirichlet_alpha),
                                float(dirichlet_epsilon), int(temperature_threshold),
                                int(pool_nodes_per_game), int(seed), self.device.index)
        h = C.c_void_p()
        self._call("az_engine_create", C.byref(h), C.byref(cfg))
        self.h = h
        self.planes = torch.zeros((self.n, 3, 8, 8), dtype=torch.float32, device=self.device)
        self.need = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
        self.probs = torch.zeros((self.n, 192), dtype=torch.float32, device=self.device)
        self.values = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        self.net = None

    def _call(self, name, *args):
        """One libazamd entry point with this engine's GPU current (kernels are launched on the
        current device; the stream passed along is that device's current stream)."""
        with torch.cuda.device(self.device):
            check(getattr(_lib.lib(), name)(*args))

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().az_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------ positions
    def set_positions(self, slots, white, black, turn, winner=None):
        slots = np.ascontiguousarray(slots, np.int32)
        white = np.ascontiguousarray(white, np.uint64)
        black = np.ascontiguousarray(black, np.uint64)
        turn = np.ascontiguousarray(turn, np.int8)
        winner = None if winner is None else np.ascontiguousarray(winner, np.int8)
        self._call("az_engine_set_positions",
            self.h, len(slots), _host_ptr(slots), _host_ptr(white), _host_ptr(black),
            _host_ptr(turn), _host_ptr(winner), _stream(self.device))

    def update_positions(self, slots, white, black, turn, winner=None):
        """Like set_positions but keeps the trees."""
        slots = np.ascontiguousarray(slots, np.int32)
        white = np.ascontiguousarray(white, np.uint64)
        black = np.ascontiguousarray(black, np.uint64)
        turn = np.ascontiguousarray(turn, np.int8)
        winner = None if winner is None else np.ascontiguousarray(winner, np.int8)
        self._call("az_engine_update_positions",
            self.h, len(slots), _host_ptr(slots), _host_ptr(white), _host_ptr(black),
            _host_ptr(turn), _host_ptr(winner), _stream(self.device))

    def root_counts(self):
        k = np.zeros(self.n, np.int32)
        self._call("az_engine_root_counts", self.h, _host_ptr(k), _stream(self.device))
        return k

    def reset_all(self):
        n = self.n
        self.set_positions(np.arange(n), np.full(n, 0xFFFF, np.uint64),
                           np.full(n, 0xFFFF << 48, np.uint64), np.ones(n, np.int8))

    def get_positions(self):
        white = np.zeros(self.n, np.uint64)
        black = np.zeros(self.n, np.uint64)
        turn = np.zeros(self.n, np.int8)
        winner = np.zeros(self.n, np.int8)
        self._call("az_engine_get_positions", self.h, _host_ptr(white), _host_ptr(black),
                                                 _host_ptr(turn), _host_ptr(winner), _stream(self.device))
        return white, black, turn, winner

    def set_active(self, active):
        a = np.ascontiguousarray(active, np.uint8)
        assert a.shape == (self.n,)
        self._call("az_engine_set_active", self.h, _host_ptr(a), _stream(self.device))

    # ------------------------------------------------------------ search steps
    def roots_prepare(self):
        self._call("az_engine_roots_prepare", self.h, _dev_ptr(self.planes), _dev_ptr(self.need), _stream(self.device))

    def roots_expand(self, probs):
        assert probs.is_cuda and probs.dtype == torch.float32 and probs.shape == (self.n, 192) and probs.is_contiguous()
        self._call("az_engine_roots_expand", self.h, _dev_ptr(probs), _stream(self.device))

    def add_noise(self, noise=None):
        """noise: [n,48] float64 Dirichlet draws (host array or cuda tensor) or None
        to draw on the device."""
        if noise is not None and not torch.is_tensor(noise):
            noise = torch.from_numpy(np.ascontiguousarray(noise, np.float64)).to(self.device)
        if noise is not None:
            assert noise.dtype == torch.float64 and noise.shape == (self.n, 48) and noise.is_contiguous()
        self._call("az_engine_add_noise", self.h, _dev_ptr(noise), _stream(self.device))

    def select(self):
        self._call("az_engine_select", self.h, _dev_ptr(self.planes), _dev_ptr(self.need), _stream(self.device))

    def expand_backup(self, probs, values):
        assert probs.is_cuda and probs.dtype == torch.float32 and probs.shape == (self.n, 192) and probs.is_contiguous()
        assert values.is_cuda and values.dtype == torch.float32 and values.shape == (self.n,) and values.is_contiguous()
        self._call("az_engine_expand_backup", self.h, _dev_ptr(probs), _dev_ptr(values), _stream(self.device))

    # ------------------------------------------------------------ queries
    def node_children(self, g, node=-1):
        k = C.c_int32()
        act = np.zeros(48, np.uint8)
        N = np.zeros(48, np.uint32)
        W = np.zeros(48, np.float32)
        P = np.zeros(48, np.float64)
        is64 = np.zeros(48, np.uint8)
        idx = np.zeros(48, np.int64)
        nN = C.c_uint32()
        nW = C.c_float()
        nidx = C.c_int64()
        self._call("az_engine_node_children",
            self.h, int(g), int(node), C.byref(k), _host_ptr(act), _host_ptr(N), _host_ptr(W),
            _host_ptr(P), _host_ptr(is64), _host_ptr(idx), C.byref(nN), C.byref(nW),
            C.byref(nidx), _stream(self.device))
        if k.value < 0:
            return None
        kk = k.value
        return dict(actions=act[:kk].copy(), N=N[:kk].copy(), W=W[:kk].copy(), P=P[:kk].copy(),
                    p_is64=is64[:kk].copy(), index=idx[:kk].copy(), root_N=int(nN.value),
                    root_W=float(nW.value), node=int(nidx.value))

    def roots_children(self, slots):
        """node_children of the roots of several games in one launch; a list of dicts
        (None for a game without a root)."""
        slots = np.ascontiguousarray(slots, np.int32)
        n = len(slots)
        k = np.zeros(n, np.int32)
        act = np.zeros((n, 48), np.uint8)
        N = np.zeros((n, 48), np.uint32)
        W = np.zeros((n, 48), np.float32)
        P = np.zeros((n, 48), np.float64)
        is64 = np.zeros((n, 48), np.uint8)
        idx = np.zeros((n, 48), np.int64)
        nN = np.zeros(n, np.uint32)
        nW = np.zeros(n, np.float32)
        nidx = np.zeros(n, np.int64)
        self._call("az_engine_roots_children",
            self.h, n, _host_ptr(slots), _host_ptr(k), _host_ptr(act), _host_ptr(N), _host_ptr(W),
            _host_ptr(P), _host_ptr(is64), _host_ptr(idx), _host_ptr(nN), _host_ptr(nW), _host_ptr(nidx),
            _stream(self.device))
        out = []
        for i in range(n):
            kk = int(k[i])
            out.append(None if kk < 0 else dict(
                actions=act[i, :kk], N=N[i, :kk], W=W[i, :kk], P=P[i, :kk], p_is64=is64[i, :kk],
                index=idx[i, :kk], root_N=int(nN[i]), root_W=float(nW[i]), node=int(nidx[i])))
        return out

    def root_children(self, g):
        return self.node_children(g, -1)

    def action_probs_all(self, temps=None, out=None):
        out = self.probs if out is None else out
        t = None if temps is None else np.ascontiguousarray(temps, np.float32)
        self._call("az_engine_action_probs", self.h, _host_ptr(t), _dev_ptr(out), _stream(self.device))
        return out

    def advance_all(self, actions):
        a = np.ascontiguousarray(actions, np.int16)
        assert a.shape == (self.n,)
        self._call("az_engine_advance", self.h, _host_ptr(a), _stream(self.device))

    def set_root(self, g, node):
        self._call("az_engine_set_root", self.h, int(g), int(node), _stream(self.device))

    def set_roots(self, slots, nodes):
        """Re-root several games in one launch (slots[i] at pool node nodes[i])."""
        slots = np.ascontiguousarray(slots, np.int32)
        nodes = np.ascontiguousarray(nodes, np.int64)
        assert len(slots) == len(nodes)
        self._call("az_engine_set_roots", self.h, len(slots), _host_ptr(slots), _host_ptr(nodes), _stream(self.device))

    def import_tree(self, g, visits, value_sum, prior, first_child, n_children, action, uniform64=None, root_prior64=None):
        """Game g's tree becomes the host-built one given as breadth-first arrays (az_engine_import_tree)."""
        n = len(visits)
        arr = [np.ascontiguousarray(visits, np.uint32), np.ascontiguousarray(value_sum, np.float32),
               np.ascontiguousarray(prior, np.float32), np.ascontiguousarray(first_child, np.uint32),
               np.ascontiguousarray(n_children, np.uint8), np.ascontiguousarray(action, np.uint8)]
        assert all(len(a) == n for a in arr)
        u = None if uniform64 is None else np.ascontiguousarray(uniform64, np.uint8)
        r = None if root_prior64 is None else np.ascontiguousarray(root_prior64, np.float64)
        self._call("az_engine_import_tree", self.h, int(g), n, *[_host_ptr(a) for a in arr], _host_ptr(u), _host_ptr(r),
                   _stream(self.device))

    def pool_usage(self):
        used = np.zeros(self.n, np.uint32)
        self._call("az_engine_pool_usage", self.h, _host_ptr(used), _stream(self.device))
        return used

    def mean_depth(self):
        d, n = C.c_uint64(), C.c_uint64()
        self._call("az_engine_stats", self.h, C.byref(d), C.byref(n), _stream(self.device))
        return d.value / max(1, n.value)

    # ------------------------------------------------------------ one search_batch
    def search_with(self, evaluate, num_simulations, add_noise=False, noise=None):
        """MCTS.search_batch (src/mcts.py:94-189) with an arbitrary evaluator between
        the device steps.  evaluate(planes_cuda[n,3,8,8], need_cuda[n]) ->
        (probs_cuda[n,192] f32, values_cuda[n] f32)."""
        self.roots_prepare()
        if bool(self.need.any().item()):
            probs, _ = evaluate(self.planes, self.need)
            self.roots_expand(probs)
        else:
            self.roots_expand(self.probs)
        if add_noise or noise is not None:
            self.add_noise(noise)
        for _ in range(num_simulations):
            self.select()
            probs, values = evaluate(self.planes, self.need)
            self.expand_backup(probs, values)

    # ------------------------------------------------------------ fused self-play
    def attach_net(self, inference):
        """inference: alphazero_amd.model.HipInference with max_batch >= n_games."""
        self._call("az_engine_attach_net", self.h, inference.h)
        self.net = inference

    def search_leaf_parallel(self, slots, num_simulations, leaf_batch):
        """num_simulations simulations of the games in `slots`, leaf_batch descents per network call
        with virtual loss (az_engine_search_leaf_parallel; results differ from search())."""
        slots = np.ascontiguousarray(slots, np.int32)
        self._call("az_engine_search_leaf_parallel", self.h, len(slots), _host_ptr(slots), int(num_simulations),
                   int(leaf_batch), _stream(self.device))

    def search(self, num_simulations, add_noise=False):
        self._call("az_engine_search", self.h, int(num_simulations), int(bool(add_noise)), _stream(self.device))

    def play_move(self, num_simulations, add_noise=True, actions=None, restart_finished=True):
        a = None if actions is None else np.ascontiguousarray(actions, np.int16)
        self._call("az_engine_play_move", self.h, int(num_simulations), int(bool(add_noise)),
                                             _host_ptr(a), int(bool(restart_finished)), _stream(self.device))

    def chosen_actions(self):
        a = np.zeros(self.n, np.int16)
        self._call("az_engine_chosen_actions", self.h, _host_ptr(a), _stream(self.device))
        return a

    def samples_pending(self):
        n = C.c_int64()
        self._call("az_engine_drain_samples", self.h, 0, None, None, None, C.byref(n), _stream(self.device))
        return n.value

    def drain_samples(self):
        """-> (states[n,3,8,8], policies[n,192], wls[n,2]) float32 cuda tensors."""
        n = self.samples_pending()
        states = torch.empty((n, 3, 8, 8), dtype=torch.float32, device=self.device)
        policies = torch.empty((n, 192), dtype=torch.float32, device=self.device)
        wls = torch.empty((n, 2), dtype=torch.float32, device=self.device)
        if n:
            got = C.c_int64()
            self._call("az_engine_drain_samples", self.h, n, _dev_ptr(states), _dev_ptr(policies),
                                                     _dev_ptr(wls), C.byref(got), _stream(self.device))
            assert got.value == n
        return states, policies, wls

    def drain_packed(self):
        """The queued samples in the wire format of the exchange between GPUs:
        a uint8 cuda tensor [n, 792] (see unpack_samples); empties the queue."""
        n = self.samples_pending()
        packed = torch.empty((n, _lib.PACKED_SAMPLE_BYTES), dtype=torch.uint8, device=self.device)
        if n:
            got = C.c_int64()
            self._call("az_engine_drain_packed", self.h, n, _dev_ptr(packed), C.byref(got), _stream(self.device))
            assert got.value == n
        return packed

    def stats(self):
        st = _lib.SelfplayStats()
        self._call("az_engine_selfplay_stats", self.h, C.byref(st), _stream(self.device))
        return dict(simulations=st.simulations, nn_evals=st.nn_evals, moves=st.moves,
                    games_finished=st.games_finished, samples_pending=st.samples_pending,
                    network_rows=st.network_rows)
