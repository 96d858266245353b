"""ctypes binding of the CPU oracle (oracle/az_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and
by the cpu_baseline leg of bench.py.  The shipped package never imports this.
Parity status: pinned against the imported reference (tests/golden).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libazoracle.so")


def build(force=False):
    src = os.path.join(_HERE, "az_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libazoracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Pos(C.Structure):
    _fields_ = [("white", C.c_uint64), ("black", C.c_uint64),
                ("turn", C.c_int32), ("winner", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER
        L.azo_start_position.argtypes = [P(Pos)]
        L.azo_legal_actions.argtypes = [P(Pos), C.c_void_p]
        L.azo_legal_actions.restype = C.c_int
        L.azo_step_action.argtypes = [P(Pos), C.c_int]
        L.azo_is_terminal.argtypes = [P(Pos)]
        L.azo_is_terminal.restype = C.c_int
        L.azo_get_result.argtypes = [P(Pos), P(C.c_float), P(C.c_float)]
        L.azo_encode_state.argtypes = [P(Pos), C.c_void_p]
        L.azo_legal_mask.argtypes = [P(Pos), C.c_void_p]
        L.azo_status.argtypes = [P(Pos)]
        L.azo_status.restype = C.c_int
        L.azo_movegen_batch.argtypes = [C.c_int64] + [C.c_void_p] * 7
        L.azo_random_positions.argtypes = [C.c_int64, C.c_uint64] + [C.c_void_p] * 4
        L.azo_mcts_create.argtypes = [C.c_int, C.c_double]
        L.azo_mcts_create.restype = C.c_void_p
        L.azo_mcts_destroy.argtypes = [C.c_void_p]
        L.azo_mcts_set_game.argtypes = [C.c_void_p, C.c_int, P(Pos)]
        L.azo_mcts_get_game.argtypes = [C.c_void_p, C.c_int, P(Pos)]
        L.azo_mcts_set_active.argtypes = [C.c_void_p, C.c_void_p]
        L.azo_mcts_roots_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.azo_mcts_roots_prepare.restype = C.c_int
        L.azo_mcts_roots_expand.argtypes = [C.c_void_p, C.c_void_p]
        L.azo_mcts_add_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        L.azo_mcts_select.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.azo_mcts_select.restype = C.c_int
        L.azo_mcts_expand_backup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.azo_mcts_root_children.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
        L.azo_mcts_root_children.restype = C.c_int
        L.azo_mcts_action_probs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.azo_mcts_advance.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.azo_mcts_mean_depth.argtypes = [C.c_void_p]
        L.azo_mcts_mean_depth.restype = C.c_double
        L.azo_mcts_total_nodes.argtypes = [C.c_void_p]
        L.azo_mcts_total_nodes.restype = C.c_int64
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---------------------------------------------------------------- game tier

def start_position():
    p = Pos()
    lib().azo_start_position(C.byref(p))
    return p


def make_pos(white, black, turn, winner=0):
    return Pos(int(white), int(black), int(turn), int(winner))


def legal_actions(p):
    buf = np.zeros(48, np.uint8)
    n = lib().azo_legal_actions(C.byref(p), _ptr(buf))
    return buf[:n].copy()


def step_action(p, action):
    q = Pos(p.white, p.black, p.turn, p.winner)
    lib().azo_step_action(C.byref(q), int(action))
    return q


def is_terminal(p):
    return bool(lib().azo_is_terminal(C.byref(p)))


def get_result(p):
    w, l = C.c_float(), C.c_float()
    lib().azo_get_result(C.byref(p), C.byref(w), C.byref(l))
    return (w.value, l.value)


def encode_state(p):
    out = np.zeros((3, 8, 8), np.float32)
    lib().azo_encode_state(C.byref(p), _ptr(out))
    return out


def legal_mask(p):
    out = np.zeros(192, np.uint8)
    lib().azo_legal_mask(C.byref(p), _ptr(out))
    return out.astype(bool)


def movegen_batch(white, black, turn, want_mask=True, want_planes=True):
    n = len(white)
    white = np.ascontiguousarray(white, np.uint64)
    black = np.ascontiguousarray(black, np.uint64)
    turn = np.ascontiguousarray(turn, np.int8)
    legal3 = np.zeros((n, 3), np.uint64)
    mask = np.zeros((n, 192), np.uint8) if want_mask else None
    planes = np.zeros((n, 3, 8, 8), np.float32) if want_planes else None
    status = np.zeros(n, np.int8)
    lib().azo_movegen_batch(n, _ptr(white), _ptr(black), _ptr(turn), _ptr(legal3),
                            _ptr(mask), _ptr(planes), _ptr(status))
    return legal3, mask, planes, status


def random_positions(n, seed):
    white = np.zeros(n, np.uint64)
    black = np.zeros(n, np.uint64)
    turn = np.zeros(n, np.int8)
    winner = np.zeros(n, np.int8)
    lib().azo_random_positions(n, seed, _ptr(white), _ptr(black), _ptr(turn), _ptr(winner))
    return white, black, turn, winner


# ---------------------------------------------------------------- tree tier

class OracleMCTS:
    """Step-driven restatement of MCTS.search_batch (mcts.py:94-189)."""

    def __init__(self, n_games, c_puct=1.5):
        self.n = n_games
        self.h = lib().azo_mcts_create(n_games, float(c_puct))
        self.planes = np.zeros((n_games, 3, 8, 8), np.float32)
        self.need = np.zeros(n_games, np.uint8)

    def __del__(self):
        if getattr(self, "h", None):
            lib().azo_mcts_destroy(self.h)
            self.h = None

    def set_game(self, g, pos):
        lib().azo_mcts_set_game(self.h, g, C.byref(pos))

    def get_game(self, g):
        p = Pos()
        lib().azo_mcts_get_game(self.h, g, C.byref(p))
        return p

    def set_active(self, active):
        a = np.ascontiguousarray(active, np.uint8)
        lib().azo_mcts_set_active(self.h, _ptr(a))

    def roots_prepare(self):
        return lib().azo_mcts_roots_prepare(self.h, _ptr(self.planes), _ptr(self.need))

    def roots_expand(self, probs):
        probs = np.ascontiguousarray(probs, np.float32)
        assert probs.shape == (self.n, 192)
        lib().azo_mcts_roots_expand(self.h, _ptr(probs))

    def add_noise(self, noise, epsilon=0.25):
        noise = np.ascontiguousarray(noise, np.float64)
        assert noise.shape == (self.n, 48)
        lib().azo_mcts_add_noise(self.h, _ptr(noise), float(epsilon))

    def select(self):
        return lib().azo_mcts_select(self.h, _ptr(self.planes), _ptr(self.need))

    def expand_backup(self, probs, values):
        probs = np.ascontiguousarray(probs, np.float32)
        values = np.ascontiguousarray(values, np.float32)
        assert probs.shape == (self.n, 192) and values.shape == (self.n,)
        lib().azo_mcts_expand_backup(self.h, _ptr(probs), _ptr(values))

    def root_children(self, g):
        act = np.zeros(48, np.uint8)
        N = np.zeros(48, np.uint32)
        W = np.zeros(48, np.float32)
        P = np.zeros(48, np.float64)
        is64 = np.zeros(48, np.uint8)
        rn = np.zeros(1, np.uint32)
        rw = np.zeros(1, np.float32)
        k = lib().azo_mcts_root_children(self.h, g, _ptr(act), _ptr(N), _ptr(W),
                                         _ptr(P), _ptr(is64), _ptr(rn), _ptr(rw))
        if k < 0:
            return None
        return dict(actions=act[:k].copy(), N=N[:k].copy(), W=W[:k].copy(),
                    P=P[:k].copy(), p_is64=is64[:k].copy(), root_N=int(rn[0]),
                    root_W=float(rw[0]))

    def action_probs(self, g, temperature):
        assert temperature in (0, 0.0, 1, 1.0)
        out = np.zeros(192, np.float32)
        lib().azo_mcts_action_probs(self.h, g, int(temperature == 0), _ptr(out))
        return out

    def advance(self, g, action):
        lib().azo_mcts_advance(self.h, g, int(action))

    def mean_depth(self):
        return lib().azo_mcts_mean_depth(self.h)

    def total_nodes(self):
        return lib().azo_mcts_total_nodes(self.h)

    def search(self, evaluator, num_simulations, noise=None, epsilon=0.25):
        """One search_batch call.  evaluator(planes[n,3,8,8], need[n]) ->
        (probs[n,192] f32, values[n] f32); rows with need == 0 are ignored."""
        if self.roots_prepare():
            probs, _ = evaluator(self.planes, self.need)
            self.roots_expand(probs)
        if noise is not None:
            self.add_noise(noise, epsilon)
        for _ in range(num_simulations):
            if self.select():
                probs, values = evaluator(self.planes, self.need)
            else:
                probs = np.zeros((self.n, 192), np.float32)
                values = np.zeros(self.n, np.float32)
            self.expand_backup(probs, values)
