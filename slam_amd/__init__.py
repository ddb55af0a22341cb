"""MI355X-native two-view initialisation: 8-point RANSAC + DLT triangulation.

The compute lives in ``slam_amd/lib/libslam_amd.so`` (hand-written HIP for
gfx950 behind a C ABI, see ``include/slam_amd.h``).  The modules here mirror the
reference's Python interface (``src/epipolar.py``, ``src/triangulation.py``)
and forward to that library through ctypes; importing this package does not
load the library, the first call does, and it raises if the library is absent.
"""
__all__ = ["epipolar", "triangulation", "util", "logger", "synth", "pipeline"]
