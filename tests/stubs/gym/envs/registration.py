import importlib

from ..error import UnregisteredEnv


class EnvSpec(object):
    def __init__(self, id, entry_point, nondeterministic=False, kwargs=None):
        self.id, self.entry_point, self.nondeterministic, self.kwargs = id, entry_point, nondeterministic, dict(kwargs or {})

    def make(self, **kwargs):
        module, _, attr = self.entry_point.partition(":")
        cls = getattr(importlib.import_module(module), attr)
        args = dict(self.kwargs)
        args.update(kwargs)
        env = cls(**args)
        env.unwrapped.spec = self
        return env


registry = {}


def register(id, entry_point=None, nondeterministic=False, kwargs=None, **_ignored):
    registry[id] = EnvSpec(id, entry_point, nondeterministic, kwargs)


def make(id, **kwargs):
    if id not in registry:
        raise UnregisteredEnv("No registered env with id: %s" % id)
    return registry[id].make(**kwargs)
