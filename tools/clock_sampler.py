#!/usr/bin/env python3
"""Samples the GPU's shader clock, power and temperature from sysfs about every millisecond until killed, one line per
sample: monotonic_ns sclk_mhz power_uW temp_mC.  Runs as a process of its own beside a measurement, which records
monotonic_ns marks of its own (tools/state_probe.py).    usage: clock_sampler.py out.txt"""
import glob
import re
import sys
import time

dev = "/sys/class/drm/card0/device"
hw = glob.glob(dev + "/hwmon/hwmon*")[0]
out = open(sys.argv[1], "w")
star = re.compile(r"(\d+)Mhz \*")
while True:
    try:
        s = star.search(open(dev + "/pp_dpm_sclk").read())
        p = open(hw + "/power1_input").read().strip()
        t = open(hw + "/temp2_input").read().strip()
    except OSError:
        continue
    out.write(f"{time.monotonic_ns()} {s.group(1) if s else -1} {p} {t}\n")
    out.flush()
    time.sleep(0.0005)
