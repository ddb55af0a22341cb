#!/usr/bin/env python3
"""Writes the bench workload's normalised matches (N=4096, 30 % outliers) as raw doubles for tools/mtbench.cpp."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, pipeline, synth
ctx = _lib.Context(0)
pair = synth.make_pair(4096, 0.3, seed=2)
n0, _ = pipeline.normalize_points(pair["pts0"], ctx=ctx)
n1, _ = pipeline.normalize_points(pair["pts1"], ctx=ctx)
with open(sys.argv[1], "wb") as f:
    f.write(np.ascontiguousarray(n0, dtype=np.float64).tobytes())
    f.write(np.ascontiguousarray(n1, dtype=np.float64).tobytes())
