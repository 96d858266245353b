"""Adapter giving the device engine the same step protocol as oracle.OracleMCTS,
so that one driver (tests/selfplay_replay.py, the golden checks) runs both."""
import numpy as np
import torch

from oracle import oracle as orc
from alphazero_amd.engine import Engine


class HostEvaluatorOnDevice:
    """Wraps a host evaluator (planes ndarray -> probs, values) for Engine.search_with."""

    def __init__(self, ev, device):
        self.ev = ev
        self.device = device

    def __call__(self, planes, need):
        p, v = self.ev(planes.cpu().numpy())
        return (torch.from_numpy(np.ascontiguousarray(p, np.float32)).to(self.device),
                torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(self.device))


class EngineBackend:
    def __init__(self, n_games, c_puct=1.5, **kw):
        self.e = Engine(n_games, c_puct=c_puct, **kw)
        self.n = n_games

    def set_game(self, g, pos):
        self.e.set_positions([g], [pos.white], [pos.black], [pos.turn], [pos.winner])

    def get_game(self, g):
        w, b, t, win = self.e.get_positions()
        return orc.make_pos(w[g], b[g], t[g], win[g])

    def set_active(self, active):
        self.e.set_active(active)

    def search(self, evaluator, num_simulations, noise=None, epsilon=0.25):
        self.e.search_with(HostEvaluatorOnDevice(evaluator, self.e.device), num_simulations,
                           noise=noise)

    def root_children(self, g):
        return self.e.root_children(g)

    def action_probs(self, g, temperature):
        temps = np.full(self.n, float(temperature), np.float32)
        return self.e.action_probs_all(temps)[g].cpu().numpy()

    def advance(self, g, action):
        a = np.full(self.n, -1, np.int16)
        a[g] = action
        self.e.advance_all(a)

    def mean_depth(self):
        return self.e.mean_depth()
