"""Accuracy of the GPU triangulation against the C restatement on 200000 synthetic matches (30 % wrong)."""
import sys

sys.path.insert(0, ".")
import numpy as np

from oracle import c_oracle as co
from oracle import slam_oracle as so
from slam_amd import synth, triangulation

d = synth.make_pair(200000, 0.3, seed=5)
K = d["K"]
P0 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
P1 = K @ so.rotation_translation_to_pose(d["R"], d["t"])[:3]
X = triangulation.triangulate_points_from_pose(P0, P1, d["pts0"], d["pts1"])
ref = co.triangulate(P0, P1, d["pts0"], d["pts1"], threads=8)
rel = np.abs(X - ref) / np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max(axis=1, keepdims=True))
r = rel.max(axis=1)
print("quantiles 50/90/99/99.9/99.99/max:", [float(f"{np.quantile(r, q):.3g}") for q in (0.5, 0.9, 0.99, 0.999, 0.9999, 1.0)])
lap = so.triangulate(P0, P1, d["pts0"][:20000], d["pts1"][:20000])
rl = (np.abs(X[:20000] - lap) / np.maximum(np.abs(lap), 1e-3 * np.abs(lap).max(axis=1, keepdims=True))).max(axis=1)
rc = (np.abs(ref[:20000] - lap) / np.maximum(np.abs(lap), 1e-3 * np.abs(lap).max(axis=1, keepdims=True))).max(axis=1)
print("vs LAPACK (20000): gpu", [float(f"{np.quantile(rl, q):.3g}") for q in (0.5, 0.99, 0.9999, 1.0)], "C restatement", [float(f"{np.quantile(rc, q):.3g}") for q in (0.5, 0.99, 0.9999, 1.0)])
