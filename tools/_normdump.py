import sys, numpy as np
from slam_amd import _lib, pipeline, synth
ctx = _lib.default_context()
outs = []
for n, seed in ((2048, 1), (4096, 2), (777, 3), (65536, 4), (9, 5)):
    d = synth.make_pair(n, 0.3, seed=seed)
    a, Ta = pipeline.normalize_points(d["pts0"], ctx=ctx)
    outs += [a, Ta]
    r = pipeline.initialise_pair(d["pts0"], d["pts1"], d["K"], iterations=200, ctx=ctx)
    outs += [r["F"], r["X"]]
np.savez(sys.argv[1], *outs)
