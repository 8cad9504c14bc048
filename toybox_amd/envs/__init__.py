from .base import AmidarEnv, BreakoutEnv, ENV_IDS, GridWorldEnv, MockALE, SpaceInvadersEnv, ToyboxBaseEnv, hash_seed, make, register_with_gym  # noqa: F401
from .constants import ACTION_LOOKUP, ACTION_MEANING  # noqa: F401
from .vec_env import ToyboxPreprocVecEnv, ToyboxVecEnv  # noqa: F401

try:                                                    # the reference registers its ids on import (toybox/__init__.py:8-24)
    import gym as _gym  # noqa: F401
except ImportError:
    REGISTERED_WITH_GYM = []
else:
    REGISTERED_WITH_GYM = register_with_gym()
