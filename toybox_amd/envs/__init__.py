from .base import AmidarEnv, BreakoutEnv, ENV_IDS, GridWorldEnv, MockALE, SpaceInvadersEnv, ToyboxBaseEnv, hash_seed, make  # noqa: F401
from .constants import ACTION_LOOKUP, ACTION_MEANING  # noqa: F401
from .vec_env import ToyboxPreprocVecEnv, ToyboxVecEnv  # noqa: F401
