"""Print-level logger with the reference's interface (src/logger.py): set_log_level, debug, info, warn, error."""
_ORDER = ("DEBUG", "INFO", "WARN", "ERROR")
LEVELS_MAP = {name: rank for rank, name in enumerate(_ORDER)}
LOG_LEVEL = LEVELS_MAP["DEBUG"]


def set_log_level(level):
    global LOG_LEVEL
    assert level in LEVELS_MAP
    LOG_LEVEL = LEVELS_MAP[level]


def _emit(rank, args):
    if LOG_LEVEL <= rank:
        print(*args)


def debug(*args):
    _emit(0, args)


def info(*args):
    _emit(1, args)


def warn(*args):
    _emit(2, args)


def error(*args):
    _emit(3, args)
