#!/usr/bin/env python3
"""How many fused-kernel grids are in flight at a time, from a rocprofv3 --kernel-trace CSV of a multi-stream run.

usage: overlap.py kernel_trace.csv [out.json]

The trace is cut into bursts at idle gaps (no kernel running for more than 100 us).  For every burst of at least 500
dispatches of the kernel: the step period (span / dispatches), the kernel's mean duration, the share of the burst's time
with 0, 1, 2, ... grids in flight, and the dispatches per hardware queue."""
import csv
import json
import sys
from collections import Counter

KERNEL = "k_ransac"
GAP_NS = 100_000


def main():
    rows = []
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            if KERNEL not in r["Kernel_Name"]:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
    rows.sort()
    bursts, cur, busy_until = [], [], None
    for s, e, q in rows:
        if busy_until is not None and s - busy_until > GAP_NS:
            bursts.append(cur)
            cur = []
        cur.append((s, e, q))
        busy_until = e if busy_until is None else max(busy_until, e)
    if cur:
        bursts.append(cur)
    out = []
    for b in bursts:
        if len(b) < 500:
            continue
        t0, t1 = b[0][0], max(e for _, e, _ in b)
        events = sorted([(s, 1) for s, _, _ in b] + [(e, -1) for _, e, _ in b])
        share, depth, last = Counter(), 0, t0
        for t, d in events:
            share[depth] += t - last
            last = t
            depth += d
        span = t1 - t0
        starts_at_depth = Counter()
        active = []
        for s, e, _ in b:     # grids already in flight when this one starts
            active = [x for x in active if x > s]
            starts_at_depth[len(active)] += 1
            active.append(e)
        out.append({"dispatches": len(b), "span_us": span / 1e3, "period_us": span / 1e3 / len(b),
                    "mean_kernel_us": sum(e - s for s, e, _ in b) / len(b) / 1e3,
                    "mean_grids_in_flight": sum(k * v for k, v in share.items()) / span,
                    "time_share_by_grids_in_flight": {str(k): round(v / span, 4) for k, v in sorted(share.items())},
                    "grids_in_flight_at_start": {str(k): v for k, v in sorted(starts_at_depth.items())},
                    "dispatches_per_queue": dict(Counter(q for _, _, q in b))})
    text = json.dumps({"kernel": KERNEL, "bursts": out}, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    for o in out:
        print(f"{o['dispatches']} dispatches, period {o['period_us']:.2f} us, kernel {o['mean_kernel_us']:.2f} us, "
              f"in flight {o['mean_grids_in_flight']:.2f}, share {o['time_share_by_grids_in_flight']}, queues {o['dispatches_per_queue']}")


if __name__ == "__main__":
    main()
