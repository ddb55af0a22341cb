"""Drop-in for the reference's src/epipolar.py: same module path, backed by slam_amd.epipolar."""
from slam_amd.epipolar import *  # noqa: F401,F403
