#!/usr/bin/env python3
"""Runs one of the tools against another build of the library (A/B runs of a compile-time variant):
    with_lib.py <library file name in slam_amd/lib> <tool.py> [the tool's options]"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from slam_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
tool = sys.argv[2]
sys.argv = [tool] + sys.argv[3:]
runpy.run_path(os.path.join(HERE, tool), run_name="__main__")
