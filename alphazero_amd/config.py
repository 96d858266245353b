"""Hyper-parameters of the self-play path.  Values and names follow the
reference's src/config.py:11-69 so that code written against it keeps working."""
import os


class Config:
    # game (src/config.py:15-18)
    BOARD_SIZE = 8
    NUM_ACTIONS = 192
    OUTPUT_ACTIONS = 192
    INPUT_PLANES = 3
    # network (src/config.py:26-28)
    RESNET_BLOCKS = 20
    RESNET_FILTERS = 128
    SE_RATIO = 8
    # search (src/config.py:33-42)
    MCTS_SIMULATIONS = 400
    MCTS_SIMULATIONS_INFERENCE = 200
    C_PUCT = 1.5
    FPU_REDUCTION = 0.0
    DIRICHLET_ALPHA = 0.35
    DIRICHLET_EPSILON = 0.25
    TEMPERATURE_THRESHOLD = 16
    # training (src/config.py:47-56)
    BATCH_SIZE = 1024
    LEARNING_RATE = 0.001
    LR_SCHEDULER_T_MAX = 200
    LR_SCHEDULER_ETA_MIN = 1e-5
    WEIGHT_DECAY = 1e-4
    GRAD_CLIP_NORM = 1.0
    PARALLEL_GAMES = 128
    SELFPLAY_BATCHES = 8
    BUFFER_SIZE = 300000
    TRAINING_EPOCHS = 1
    # paths (src/config.py:61-69)
    CHECKPOINT_DIR = "checkpoints"
    BEST_MODEL = "model_best.pt"
    DATA_FILE = "training_data.npz"
    ARENA_STATE = "arena_state.json"

    @classmethod
    def get_checkpoint_path(cls, filename):
        return os.path.join(cls.CHECKPOINT_DIR, filename)
