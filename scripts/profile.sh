#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  bash scripts/profile.sh r01
# Outputs under gpurun_out/prof_<tag>/ ; condense them with scripts/profile_summary.py into profiles/.
#  1. rocprofv3 --kernel-trace --stats of the default bench command (and of the n=21 variant)
#  2. separate --pmc passes (FETCH_SIZE, then WRITE_SIZE; never combined with trace domains) of the bench command,
#     the cache-busting 1M-filter run, the n=21 run, and the calibration copy (known bytes, same access pattern)
set -o pipefail
TAG=${1:-r01}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace64k -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/trace64k.json 2> $OUT/trace64k.err || exit 11
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace64k_n21 -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --n-states 21 > $OUT/trace64k_n21.json 2> $OUT/trace64k_n21.err || exit 11
echo "traces done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc64k_$C -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc64k_$C.json 2> $OUT/pmc64k_$C.err || exit 12
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc64k_n21_$C -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --n-states 21 > $OUT/pmc64k_n21_$C.json 2> $OUT/pmc64k_n21_$C.err || exit 12
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc1m_$C -- python3 bench.py --batch-per-gpu 1048576 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc1m_$C.json 2> $OUT/pmc1m_$C.err || exit 13
  rocprofv3 --pmc $C --output-format csv -d $OUT/calib_$C -- python3 scripts/calib_copy.py > $OUT/calib_$C.txt 2> $OUT/calib_$C.err || exit 14
  echo "pmc $C done"
done
python3 bench.py --batch-per-gpu 1048576 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench1m.json 2> $OUT/bench1m.err
python3 scripts/calib_copy.py > $OUT/calib_plain.txt 2>&1
echo "profile done"
