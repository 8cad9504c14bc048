from . import time_limit  # noqa: F401
from .time_limit import TimeLimit  # noqa: F401
