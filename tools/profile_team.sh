#!/bin/bash
# Counter passes over the persistent batch-1 team kernels, run on the GPU box from the repo root:
#   tools/profile_team.sh <tag>   (raw output under gpurun_out/prof_<tag>_team/, summary gpurun_out/profiles_<tag>/<tag>_team_pmc.csv)
# One --pmc set per run, never combined with a trace; the program comes directly after `--`.
set -o pipefail
tag=${1:-r3}
out=gpurun_out/prof_${tag}_team
mkdir -p $out
export TMPDIR=/tmp
run() { echo "== $*"; timeout -s KILL 240 "$@" > $out/last.log 2>&1 || { echo "FAILED: $*"; tail -5 $out/last.log; exit 1; }; tail -1 $out/last.log; }
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY"
for case in "t2 100 800" "t2 1000 2000" "t1 100 200"; do
  name=$(echo $case | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    run rocprofv3 --pmc $c -f csv -d $out/${name}_$c -o pmc -- python3 tools/team_run.py $case
  done
  run rocprofv3 --pmc $SQ -f csv -d $out/${name}_SQ -o pmc -- python3 tools/team_run.py $case
  run rocprofv3 --kernel-trace --stats -f csv -d $out/${name}_kt -o kt -- python3 tools/team_run.py $case
done
PROFILE_DST=gpurun_out/profiles_$tag python3 tools/team_pmc_collect.py $tag && rm -rf $out/t?_*
echo done
