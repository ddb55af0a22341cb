"""Stand-in for the reference's 2D plotting module (src/plot.py).  Visualisation is outside this package's scope:
every function the reference's modules and driver call exists with the reference's signature and does nothing
(one WARN line per process says so), so a driver written against the reference runs unchanged."""
from slam_amd import logger as _logger

_told = set()


def _skipped(name):
    if name not in _told:
        _told.add(name)
        _logger.warn("src.plot." + name + ": plotting is not part of this package; call ignored")


def plot_image_keypoints(image, image_keypoints):        # reference: src/plot.py:6
    _skipped("plot_image_keypoints")


def plot_image_matches(image1, image1_points, image2, image2_points, subplots=None, show=True):  # src/plot.py:13
    _skipped("plot_image_matches")


def plot_vectors(vector_list):                             # reference: src/plot.py:36
    _skipped("plot_vectors")


def plot_vectors_with_starting_point(vector_map):          # reference: src/plot.py:41
    _skipped("plot_vectors_with_starting_point")
