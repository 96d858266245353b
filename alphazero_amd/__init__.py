"""MI355X-native self-play engine for AlphaZero Breakthrough.

Drop-in counterparts of the reference's src/game.py, src/mcts.py and
src/model.py live in alphazero_amd.game / .mcts / .model; the compute runs in
libazamd.so (hand-written gfx950 HIP kernels behind the C ABI of
include/azamd.h).
"""
__all__ = ["config", "game", "mcts", "model", "engine", "selfplay"]
