"""Stand-in for the reference's Pangolin 3D viewer module (src/display.py).  The viewer is outside this package's
scope: every function the reference's driver calls exists with the reference's signature and does nothing (one
WARN line per process says so); should_quit() reports True so that the driver's viewer loop ends at once."""
from slam_amd import logger as _logger

_told = set()


def _skipped(name):
    if name not in _told:
        _told.add(name)
        _logger.warn("src.display." + name + ": the 3D viewer is not part of this package; call ignored")


def setup():                                   # reference: src/display.py:18
    _skipped("setup")


def should_quit():                             # reference: src/display.py:40
    _skipped("should_quit")
    return True


def draw_points(points):                       # reference: src/display.py:44
    _skipped("draw_points")


def draw_camera(pose, color=(0.0, 1.0, 0.0)):  # reference: src/display.py:50
    _skipped("draw_camera")


def init_frame():                              # reference: src/display.py:73
    _skipped("init_frame")


def finish_frame():                            # reference: src/display.py:80
    _skipped("finish_frame")
