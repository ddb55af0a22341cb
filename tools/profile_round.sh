#!/bin/bash
# Everything profiles/ holds for one state of the code, from one GPU box in one call:
#   tools/profile_round.sh TAG      (run from the repository root; writes gpurun_out/prof_TAG/)
# 1. rocprofv3 --kernel-trace --stats of a one-stream run of the headline step
# 2. bench lines without a profiler (default length, the short 20-step run, 2000-step bursts), after step 3
# 3. counter passes, each in a run of its own: FETCH_SIZE, WRITE_SIZE (headline step, the large-H call on either scoring
#    kernel, the triangulation), SQ counters
set -o pipefail
TAG=${1:-dev}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from slam_amd import _lib; print(_lib.build_hash())" > $OUT/build_hash.txt || exit 1
cd /tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --lanes 1 --no-aux --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err || exit 1
HEAD="python3 $ROOT/bench.py --steps 50 --warmup 5 --bursts 4 --lanes 1 --no-cpu-baseline --no-aux"
BIG="python3 $ROOT/tools/kbench.py -N 4096 -H 65536 --sampler 1 --steps 10 --f64"
BIG32="python3 $ROOT/tools/kbench.py -N 4096 -H 65536 --sampler 1 --steps 10"
TRI="python3 $ROOT/tools/tribench.py -n 16777216 --steps 3"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_head -- $HEAD > /dev/null 2> $OUT/pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_head -- $HEAD > /dev/null 2>> $OUT/pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_big -- $BIG > /dev/null 2>> $OUT/pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_big -- $BIG > /dev/null 2>> $OUT/pmc.err || exit 1
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
timeout -s KILL 120 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_sq_big -- $BIG > /dev/null 2>> $OUT/pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_big32 -- $BIG32 > /dev/null 2>> $OUT/pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_big32 -- $BIG32 > /dev/null 2>> $OUT/pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_sq_big32 -- $BIG32 > /dev/null 2>> $OUT/pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_sq_head -- $HEAD > /dev/null 2>> $OUT/pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_sq_tri -- $TRI > /dev/null 2>> $OUT/pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_tri -- $TRI > /dev/null 2>> $OUT/pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_tri -- $TRI > /dev/null 2>> $OUT/pmc.err || exit 1
cd $ROOT
for d in pmc_fetch_head pmc_write_head pmc_fetch_big pmc_write_big pmc_sq_big pmc_fetch_big32 pmc_write_big32 pmc_sq_big32 pmc_sq_head pmc_sq_tri pmc_fetch_tri pmc_write_tri; do
  python3 tools/pmc_summary.py $OUT/$d > $OUT/$d.json
  rm -rf $OUT/$d
done
# the counters of this build become profiles/traffic.json before the bench lines are written, so that the lines quote them
python3 tools/collect_profiles.py $TAG --traffic-only > /dev/null || exit 1
cp profiles/traffic.json $OUT/traffic.json
cd /tmp
timeout -k 10 300 python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 120 python3 $ROOT/bench.py --steps 20 --warmup 5 --no-aux --no-cpu-baseline > $OUT/bench_20steps.json 2>> $OUT/bench.err || exit 1
timeout -k 10 120 python3 $ROOT/bench.py --steps 2000 --warmup 200 --no-aux --no-cpu-baseline > $OUT/bench_2000steps.json 2>> $OUT/bench.err || exit 1
cd $ROOT
# 4. kernel trace of the headline run with four calls in flight (start and end of every dispatch, per hardware queue)
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace4 -- python3 $ROOT/bench.py --steps 2000 --warmup 200 --bursts 8 --no-aux --no-cpu-baseline > $OUT/trace4_bench.json 2> $OUT/trace4.err || exit 1
cd $ROOT
find $OUT/trace4 -name "*kernel_trace.csv" -exec python3 tools/overlap.py {} $OUT/overlap.json \; > $OUT/overlap.txt
rm -rf $OUT/trace4
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
ls $OUT
