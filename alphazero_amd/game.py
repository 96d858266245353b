"""Breakthrough rules with the call surface of the reference's src/game.py.

`BreakthroughGame` is the host-side game object that drivers, tests and the
search API hand around (a mutable int8 8x8 `board`, `turn`, `_winner`).  It is
bookkeeping for ONE game on the host; anything batched — move generation,
terminal detection and encoding for many positions, and every position the
search visits — runs in libazamd on the GPU (`movegen_encode` below and the
engine).  Its single-position methods are written on bitboards in the same
formulation the kernels use, so that both sides share one definition of the
rules; they are checked against the reference-generated fixtures.
"""
from typing import List, Optional, Tuple

import numpy as np

from .config import Config

BOARD_SIZE = Config.BOARD_SIZE
WHITE = 1    # moves first, starts on rows 0-1, moves towards higher rows (src/game.py:29)
BLACK = -1   # starts on rows 6-7 (src/game.py:30)
EMPTY = 0
NUM_ACTIONS = Config.NUM_ACTIONS

# (row_delta, col_delta) from WHITE's point of view (src/game.py:41-45)
DIRECTIONS = [(1, 0), (1, -1), (1, 1)]

_FILE_A = 0x0101010101010101
_FILE_H = 0x8080808080808080
_ALL = (1 << 64) - 1
_DELTA = (8, 7, 9)  # canonical target square offset per direction


def _encode_action(from_row: int, from_col: int, direction: int) -> int:
    """src/game.py:48-51"""
    return (from_row * BOARD_SIZE + from_col) * 3 + direction


def _decode_action(action: int) -> Tuple[int, int, int]:
    """src/game.py:54-60"""
    square, direction = divmod(action, 3)
    return square // BOARD_SIZE, square % BOARD_SIZE, direction


def _rev64(x: int) -> int:
    return int(f"{x:064b}"[::-1], 2)


def board_to_bits(board: np.ndarray) -> Tuple[int, int]:
    """bit (row*8+col) of (white, black) — tests/test_compatibility.py:8-17"""
    flat = np.asarray(board).reshape(64)
    w = b = 0
    for sq in np.nonzero(flat == WHITE)[0]:
        w |= 1 << int(sq)
    for sq in np.nonzero(flat == BLACK)[0]:
        b |= 1 << int(sq)
    return w, b


def bits_to_board(white: int, black: int) -> np.ndarray:
    board = np.zeros(64, dtype=np.int8)
    for sq in range(64):
        if (white >> sq) & 1:
            board[sq] = WHITE
        elif (black >> sq) & 1:
            board[sq] = BLACK
    return board.reshape(8, 8)


def canonical_bits(white: int, black: int, turn: int) -> Tuple[int, int]:
    """(mine, theirs) as seen by the side to move: Black sees the board rotated by
    180 degrees (src/game.py:299-302) = a 64-bit reversal with colours swapped."""
    if turn == WHITE:
        return white, black
    return _rev64(black), _rev64(white)


def legal_sources(mine: int, theirs: int) -> Tuple[int, int, int]:
    """Canonical source squares with a legal forward / diagonal-left / diagonal-right
    move (src/game.py:141-148): forward needs an empty target, a diagonal a target
    that is not one of the mover's own pieces."""
    empty = ~(mine | theirs) & _ALL
    fwd = mine & (empty >> 8)
    dl = mine & ~_FILE_A & ((~mine & _ALL) >> 7)
    dr = mine & ~_FILE_H & ((~mine & _ALL) >> 9)
    return fwd, dl, dr


class BreakthroughGame:
    """Drop-in for src/game.py:63-328."""

    INPUT_PLANES = Config.INPUT_PLANES

    def __init__(self, board: Optional[np.ndarray] = None, turn: int = WHITE):
        self.board = board.copy() if board is not None else self._initial_board()
        self.turn = turn
        self._winner: Optional[int] = None

    def _initial_board(self) -> np.ndarray:
        board = np.zeros((BOARD_SIZE, BOARD_SIZE), dtype=np.int8)
        board[0:2, :] = WHITE
        board[6:8, :] = BLACK
        return board

    def clone(self) -> "BreakthroughGame":
        g = BreakthroughGame.__new__(BreakthroughGame)
        g.board = self.board.copy()
        g.turn = self.turn
        g._winner = self._winner
        return g

    # -- rules -----------------------------------------------------------------
    def _canonical(self) -> Tuple[int, int]:
        w, b = board_to_bits(self.board)
        return canonical_bits(w, b, self.turn)

    def _ordered_actions(self) -> List[int]:
        """Legal actions in the reference's insertion order (src/game.py:117-118 scans
        the real board row-major): ascending canonical squares for White,
        descending for Black, directions 0,1,2 within a square."""
        srcs = legal_sources(*self._canonical())
        squares = range(64) if self.turn == WHITE else range(63, -1, -1)
        return [sq * 3 + d for sq in squares for d in range(3) if (srcs[d] >> sq) & 1]

    def get_legal_moves(self) -> List[Tuple[int, int, int, int]]:
        """src/game.py:109-150"""
        return [self.decode_action(a) for a in self._ordered_actions()]

    def _is_valid_square(self, row: int, col: int) -> bool:
        return 0 <= row < BOARD_SIZE and 0 <= col < BOARD_SIZE

    def step(self, move: Tuple[int, int, int, int]):
        """src/game.py:156-173 (no legality check, like the reference)"""
        fr, fc, tr, tc = move
        self.board[tr, tc] = self.board[fr, fc]
        self.board[fr, fc] = EMPTY
        self._check_winner(tr)
        self.turn = -self.turn

    def _check_winner(self, to_row: int):
        """src/game.py:175-187"""
        if self.turn == WHITE and to_row == BOARD_SIZE - 1:
            self._winner = WHITE
        elif self.turn == BLACK and to_row == 0:
            self._winner = BLACK
        if self._winner is None and not np.any(self.board == -self.turn):
            self._winner = self.turn

    def is_terminal(self) -> bool:
        """src/game.py:189-194"""
        if self._winner is not None:
            return True
        fwd, dl, dr = legal_sources(*self._canonical())
        return (fwd | dl | dr) == 0

    def get_result(self) -> Tuple[float, float]:
        """(win, loss) from WHITE's point of view — src/game.py:196-215"""
        if not self.is_terminal():
            return (0.0, 0.0)
        if self._winner == WHITE:
            return (1.0, 0.0)
        if self._winner == BLACK:
            return (0.0, 1.0)
        return (0.0, 1.0) if self.turn == WHITE else (1.0, 0.0)

    def get_reward(self) -> float:
        w, l = self.get_result()
        return w - l

    # -- action mapping ----------------------------------------------------------
    def encode_action(self, move: Tuple[int, int, int, int]) -> int:
        """src/game.py:225-255"""
        fr, fc, tr, tc = move
        if self.turn == WHITE:
            dc = tc - fc
        else:
            fr, fc = BOARD_SIZE - 1 - fr, BOARD_SIZE - 1 - fc
            dc = (BOARD_SIZE - 1 - tc) - fc
        direction = 0 if dc == 0 else (1 if dc == -1 else 2)
        return _encode_action(fr, fc, direction)

    def decode_action(self, action: int) -> Tuple[int, int, int, int]:
        """src/game.py:257-278"""
        fr, fc, direction = _decode_action(action)
        dr, dc = DIRECTIONS[direction]
        tr, tc = fr + dr, fc + dc
        if self.turn == BLACK:
            fr, fc, tr, tc = (BOARD_SIZE - 1 - fr, BOARD_SIZE - 1 - fc,
                              BOARD_SIZE - 1 - tr, BOARD_SIZE - 1 - tc)
        return (fr, fc, tr, tc)

    # -- encodings ---------------------------------------------------------------
    def get_encoded_state(self) -> np.ndarray:
        """float32 [3,8,8]: mine, theirs, ones — src/game.py:280-307"""
        planes = np.zeros((self.INPUT_PLANES, BOARD_SIZE, BOARD_SIZE), dtype=np.float32)
        board = self.board if self.turn == WHITE else self.board[::-1, ::-1]
        planes[0] = board == self.turn
        planes[1] = board == -self.turn
        planes[2] = 1.0
        return planes

    def get_legal_action_mask(self) -> np.ndarray:
        """src/game.py:309-315"""
        mask = np.zeros(NUM_ACTIONS, dtype=np.bool_)
        mask[self._ordered_actions()] = True
        return mask

    def __str__(self) -> str:
        symbols = {WHITE: "○", BLACK: "●", EMPTY: "."}
        lines = ["  a b c d e f g h"]
        for row in range(BOARD_SIZE - 1, -1, -1):
            lines.append(f"{row + 1} " + "".join(symbols[int(self.board[row, c])] + " " for c in range(BOARD_SIZE)))
        lines.append(f"Turn: {'White' if self.turn == WHITE else 'Black'}")
        return "\n".join(lines)


Game = BreakthroughGame  # src/game.py:332


def movegen_encode(white, black, turn, want_mask=True, want_planes=True, device=None):
    """Batched GPU move generation / terminal detection / encoding
    (az_movegen_encode in include/azamd.h).

    white, black: uint64 arrays [n]; turn: int8 [n].  Returns a dict of cuda
    tensors: legal3 [n,3] (int64 bit patterns), mask [n,192] uint8, planes
    [n,3,8,8] float32, status [n] int8.
    """
    import ctypes as C

    import torch

    from . import _lib
    if not torch.cuda.is_available():
        raise _lib.AzError("movegen_encode needs a GPU (no CPU fallback exists)")
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    n = len(white)
    tw = torch.from_numpy(np.ascontiguousarray(white, np.uint64).view(np.int64)).to(dev)
    tb = torch.from_numpy(np.ascontiguousarray(black, np.uint64).view(np.int64)).to(dev)
    tt = torch.from_numpy(np.ascontiguousarray(turn, np.int8)).to(dev)
    out = dict(legal3=torch.empty((n, 3), dtype=torch.int64, device=dev),
               mask=torch.empty((n, 192), dtype=torch.uint8, device=dev) if want_mask else None,
               planes=torch.empty((n, 3, 8, 8), dtype=torch.float32, device=dev) if want_planes else None,
               status=torch.empty(n, dtype=torch.int8, device=dev))
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().az_movegen_encode(
            n, p(tw), p(tb), p(tt), p(out["legal3"]), p(out["mask"]), p(out["planes"]), p(out["status"]),
            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out
