"""Minimal logging surface used by the hot path: ``printlog`` and ``Logger.info``
(reference: utils/logger.py:31-188).  Rank-0 printing, optional file handler."""
import logging
import sys

_logger = logging.getLogger("mscs_amd")
if not _logger.handlers:
    _h = logging.StreamHandler(sys.stdout)
    _h.setFormatter(logging.Formatter("%(asctime)s %(levelname)-7s %(message)s"))
    _logger.addHandler(_h)
    _logger.setLevel(logging.INFO)
    _logger.propagate = False


class Logger:
    @staticmethod
    def init(logfile_level="info", log_file=None, stdout_level="info", rewrite=False):
        if log_file:
            fh = logging.FileHandler(log_file, mode="w" if rewrite else "a")
            fh.setFormatter(logging.Formatter("%(asctime)s %(levelname)-7s %(message)s"))
            _logger.addHandler(fh)

    @staticmethod
    def info(msg):
        _logger.info(msg)

    @staticmethod
    def warn(msg):
        _logger.warning(msg)

    @staticmethod
    def error(msg):
        _logger.error(msg)


def printlog(*args, **kwargs):
    from .distributed import get_rank
    if get_rank() == 0:
        _logger.info(" ".join(str(a) for a in args))


def set_verbosity(level):
    _logger.setLevel(level)
