#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun; outputs under gpurun_out/prof_r01/).
#  1. rocprofv3 --kernel-trace --stats of the default bench command  -> per-kernel average duration
#  2. separate --pmc passes (FETCH_SIZE, then WRITE_SIZE) of the same command and of the cache-busting 1M-filter
#     run, plus the calibration copy (known bytes, same 8 B/lane access pattern)
set -o pipefail
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/prof_r01
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-/root/repo}
BENCH="bench.py --steps 200 --warmup 20 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace64k -- python3 $BENCH > $OUT/trace64k.json 2> $OUT/trace64k.err || exit 11
echo "trace done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc64k_$C -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc64k_$C.json 2> $OUT/pmc64k_$C.err || exit 12
  echo "pmc 64k $C done"
done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc1m_$C -- python3 bench.py --batch-per-gpu 1048576 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc1m_$C.json 2> $OUT/pmc1m_$C.err || exit 13
  echo "pmc 1M $C done"
done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/calib_$C -- python3 scripts/calib_copy.py > $OUT/calib_$C.txt 2> $OUT/calib_$C.err || exit 14
  echo "calib $C done"
done
python3 bench.py --batch-per-gpu 1048576 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench1m.json 2> $OUT/bench1m.err
python3 scripts/calib_copy.py > $OUT/calib_plain.txt 2>&1
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
