import logging

_log = logging.getLogger("gym-stub")
INFO = logging.INFO


def info(msg, *args):
    _log.info(msg, *args)


def warn(msg, *args):
    _log.warning(msg, *args)
