"""Drop-in for the reference's src/logger.py: same module path, backed by slam_amd.logger."""
from slam_amd.logger import *  # noqa: F401,F403
