"""Minimal stand-in for the `gym` package (see ../README.md).  Not gym's code."""
from . import error, logger, spaces, utils, wrappers  # noqa: F401
from .core import Env, ObservationWrapper, RewardWrapper, Wrapper  # noqa: F401
from .envs.registration import make, register, registry  # noqa: F401

__version__ = "0.0-stub"
