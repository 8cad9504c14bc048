from ...core import Env


class AtariEnv(Env):
    """Name only: subclasses (the Toybox envs) bring their own step / reset / seed."""
