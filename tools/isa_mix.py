#!/usr/bin/env python3
"""Instruction mix of kernels in a gfx950 assembly listing (development aid).
usage: isa_mix.py file.s substring [substring ...]"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
# a kernel body runs from its label to the matching .Lfunc_end
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not any(k in name for k in sys.argv[2:]):
        continue
    lines = [l.strip() for l in body.split("\n")]
    lines = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    c = Counter(l.split()[0] for l in lines)
    print(name[:70], "static instructions:", len(lines))
    print("   ", ", ".join(f"{k} {v}" for k, v in c.most_common(26)))
