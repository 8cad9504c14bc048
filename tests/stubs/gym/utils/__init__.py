from . import seeding  # noqa: F401


class EzPickle(object):
    def __init__(self, *args, **kwargs):
        self._ezpickle_args = args
        self._ezpickle_kwargs = kwargs
