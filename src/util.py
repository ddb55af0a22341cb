"""Drop-in for the reference's src/util.py: same module path, backed by slam_amd.util."""
from slam_amd.util import *  # noqa: F401,F403
