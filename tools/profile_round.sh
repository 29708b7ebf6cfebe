#!/bin/bash
# rocprofv3 passes behind the numbers of a round, run on the GPU box from the repo root:
#   tools/profile_round.sh <tag>        (results under gpurun_out/prof_<tag>/; tools/profile_collect.py turns them
#                                        into the summaries kept under profiles/)
# Kernel trace + stats of the default bench command, then counter passes (one --pmc set per run, never combined
# with a trace) on the decoder alone (64 x 128, 64 steps) and on Griffin-Lim alone (64 x 800 frames).
set -o pipefail
tag=${1:-r2}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
run() { echo "== $*"; timeout -s KILL 300 "$@" > $out/last.log 2>&1 || { echo "FAILED: $*"; tail -5 $out/last.log; exit 1; }; }
run rocprofv3 --kernel-trace --stats -f csv -d $out/bench -o bench -- python3 bench.py --no-cpu-baseline --steps 2
cp $out/last.log $out/bench_under_rocprof.log
for c in FETCH_SIZE WRITE_SIZE; do
  run rocprofv3 --pmc $c -f csv -d $out/dec_$c -o pmc -- python3 tools/dec_only.py 64 128 64
done
run rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -f csv -d $out/dec_SQ -o pmc -- python3 tools/dec_only.py 64 128 64
for c in FETCH_SIZE WRITE_SIZE; do
  run rocprofv3 --pmc $c -f csv -d $out/gl_$c -o pmc -- python3 tools/gl_bench.py 64 800 1
done
run rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS -f csv -d $out/gl_SQ -o pmc -- python3 tools/gl_bench.py 64 800 1
run rocprofv3 --kernel-trace --stats -f csv -d $out/t1 -o t1 -- python3 tools/t1_bench.py
# summaries only travel back (the raw traces are tens of MB)
PROFILE_DST=gpurun_out/profiles_$tag python3 tools/profile_collect.py $tag && rm -rf $out/bench $out/t1 $out/dec_* $out/gl_*
echo done
