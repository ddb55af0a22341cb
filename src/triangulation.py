"""Drop-in for the reference's src/triangulation.py: same module path, backed by slam_amd.triangulation."""
from slam_amd.triangulation import *  # noqa: F401,F403
