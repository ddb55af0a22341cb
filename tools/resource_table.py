#!/usr/bin/env python3
"""Table of every kernel's registers, scratch, LDS and occupancy from the compiler's resource-usage remarks.
usage: make -C slam_amd/csrc resource-usage 2>&1 | python3 tools/resource_table.py > profiles/TAG_resource_usage.txt"""
import re
import subprocess
import sys

rows, cur = [], {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                  r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    key, val = m.group(1), m.group(2)
    if key == "Function Name":
        if cur:
            rows.append(cur)
        cur = {"name": val}
    else:
        cur[key] = val
if cur:
    rows.append(cur)
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("kernel | VGPRs | SGPRs | scratch B/lane | VGPRs spilled | SGPRs spilled | LDS B/workgroup | waves/SIMD")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("slamk::", "").replace("slamx::", "")
    print(f"{n} | {r.get('VGPRs')} | {r.get('TotalSGPRs')} | {r.get('ScratchSize [bytes/lane]')} | {r.get('VGPRs Spill')} | "
          f"{r.get('SGPRs Spill')} | {r.get('LDS Size [bytes/block]')} | {r.get('Occupancy [waves/SIMD]')}")
