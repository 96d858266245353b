"""Replay of ParallelTrainer.execute_parallel_episodes (parallel_trainer.py:137-223)
over any step-driven search backend, with the random draws injected."""
import numpy as np

from oracle import oracle as orc


def replay_selfplay(backend, n_games, sims, dirichlet, choices, evaluator,
                    temperature_threshold=16):
    """backend: OracleMCTS-like object (set_game/set_active/search/action_probs/
    advance/get_game).  Returns (states, policies, wls) in the reference's order."""
    for g in range(n_games):
        backend.set_game(g, orc.start_position())
    game_data = [[] for _ in range(n_games)]
    step_counts = [0] * n_games
    active = list(range(n_games))
    out_s, out_p, out_w = [], [], []
    di = ci = 0
    while active:
        mask = np.zeros(n_games, np.uint8)
        mask[active] = 1
        backend.set_active(mask)
        noise = np.zeros((n_games, 48))
        for i in active:                       # one draw per active root, in order
            noise[i] = dirichlet[di]
            di += 1
        backend.search(evaluator, sims, noise=noise)
        finished = []
        for i in active:
            temp = 1.0 if step_counts[i] < temperature_threshold else 0.0
            probs = backend.action_probs(i, temp)
            pos = backend.get_game(i)
            game_data[i].append((orc.encode_state(pos), probs, pos.turn))
            a = int(choices[ci])
            ci += 1
            backend.advance(i, a)
            step_counts[i] += 1
            pos = backend.get_game(i)
            if orc.is_terminal(pos):
                w, l = orc.get_result(pos)
                for s, p, turn in game_data[i]:
                    wl = np.array([w, l] if turn == 1 else [l, w], np.float32)
                    out_s.append(s); out_p.append(p); out_w.append(wl)
                finished.append(i)
        for i in finished:
            active.remove(i)
    assert di == len(dirichlet) and ci == len(choices)
    return out_s, out_p, out_w


def replay_selfplay_oracle(n_games, sims, dirichlet, choices, evaluator):
    return replay_selfplay(orc.OracleMCTS(n_games, 1.5), n_games, sims, dirichlet,
                           choices, evaluator)
