#!/bin/bash
# A/B of two builds of the library on the same box in one call: tools/ab.sh <libA> <libB> <kbench options...>
# (alternates A B A B so that a drift of the clocks shows up as a difference between the two runs of one build)
A=$1; B=$2; shift 2
for lib in $A $B $A $B; do
  printf "%-24s " $lib
  timeout -k 10 120 python3 tools/with_lib.py $lib kbench.py "$@" 2>&1 | tail -1
done
