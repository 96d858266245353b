"""Device-resident counterpart of ParallelTrainer.execute_parallel_episodes
(src/parallel_trainer.py:137-223)."""
import numpy as np
import torch

from .config import Config
from .engine import Engine
from .model import HipInference


class SelfPlay:
    """n_games concurrent games on one GPU: search, move choice, tree reuse and
    sample collection all run inside libazamd; the host only counts moves."""

    def __init__(self, model, n_games=Config.PARALLEL_GAMES, num_simulations=Config.MCTS_SIMULATIONS,
                 c_puct=Config.C_PUCT, seed=0, pool_nodes_per_game=0, device=None):
        self.num_simulations = int(num_simulations)
        self.inference = model if isinstance(model, HipInference) else HipInference(
            model, max_batch=max(16, n_games), device=device)
        self.engine = Engine(n_games, c_puct=c_puct, dirichlet_alpha=Config.DIRICHLET_ALPHA,
                             dirichlet_epsilon=Config.DIRICHLET_EPSILON,
                             temperature_threshold=Config.TEMPERATURE_THRESHOLD,
                             pool_nodes_per_game=pool_nodes_per_game, seed=seed, device=device)
        self.engine.attach_net(self.inference)
        self.engine.reset_all()

    @staticmethod
    def _collected(out, device_output):
        parts = tuple(torch.cat(o) for o in out)
        return parts if device_output else tuple(t.cpu().numpy() for t in parts)

    def execute_parallel_episodes(self, num_games=None, device_output=False):
        """Play every game to the end (finished games leave the batch, as in the
        reference) and return the examples as float32 arrays
        (states[n,3,8,8], policies[n,192], wls[n,2]): numpy arrays, or with
        device_output=True the cuda tensors the engine drained its queue into (no
        host copy; what a trainer on the same GPU wants)."""
        e = self.engine
        n = e.n if num_games is None else int(num_games)
        assert n <= e.n
        e.reset_all()
        self._continuous = False
        active = np.zeros(e.n, np.uint8)
        active[:n] = 1
        e.set_active(active)
        out = [[], [], []]
        while e.stats()["games_finished"] - getattr(self, "_finished0", 0) < n:
            e.play_move(self.num_simulations, add_noise=True, restart_finished=False)
            if e.samples_pending() > e.n * 64:
                for o, t in zip(out, e.drain_samples()):
                    o.append(t)
        for o, t in zip(out, e.drain_samples()):
            o.append(t)
        self._finished0 = e.stats()["games_finished"]
        return self._collected(out, device_output)

    def collect_games(self, num_games=None, device_output=False):
        """Continuous self-play: every slot of the engine stays busy (a finished game restarts
        from the start position at once) until num_games more games have finished; returns
        their examples like execute_parallel_episodes.  Games still running when the call
        returns stay in the engine and go on in the next call, so no launch ever runs below
        the engine's full batch - the reference's loop, whose batch shrinks as games finish,
        has no counterpart for this; it is the faster way to fill a replay buffer."""
        e = self.engine
        n = e.n if num_games is None else int(num_games)
        if not getattr(self, "_continuous", False):
            e.reset_all()
            e.set_active(np.ones(e.n, np.uint8))
            self._continuous = True
        start = e.stats()["games_finished"]
        out = [[], [], []]
        while e.stats()["games_finished"] - start < n:
            e.play_move(self.num_simulations, add_noise=True, restart_finished=True)
            if e.samples_pending() > e.n * 64:
                for o, t in zip(out, e.drain_samples()):
                    o.append(t)
        for o, t in zip(out, e.drain_samples()):
            o.append(t)
        self._finished0 = e.stats()["games_finished"]
        return self._collected(out, device_output)

    def step(self, restart_finished=True):
        """One ply for every active game (continuous self-play when finished games restart)."""
        self.engine.play_move(self.num_simulations, add_noise=True, restart_finished=restart_finished)
