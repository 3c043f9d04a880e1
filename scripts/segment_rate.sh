#!/bin/bash
# Recorded-log replay rates (run on the GPU box):  bash scripts/segment_rate.sh [out-file]
# Builds tests/cpp/test_segments.cpp with -O2 and times SegmentStreamer (segment_stream.hpp) on windows of 16 long recordings, the
# per-message SegmentBatcher beside it; every line carries its stage breakdown.  (/dev/shm holds the logs when it exists.)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/segment_rate.txt}
cd $ROOT
python3 -c "from oracle import po; po.build()" || exit 1
EXE=tests/build/test_segments_rate
g++ -O2 -std=c++17 -fopenmp -Wall -o $EXE tests/cpp/test_segments.cpp -Lpronto_amd/lib -lpronto_batch -Loracle/build -lpronto_oracle \
  -L/opt/rocm/lib -Wl,-rpath,$ROOT/pronto_amd/lib -Wl,-rpath,$ROOT/oracle/build -Wl,-rpath,/opt/rocm/lib || exit 2
DIR=/tmp
[ -d /dev/shm ] && DIR=/dev/shm
: > $OUT
run() { echo "\$ $*" >> $OUT; "$@" 2>&1 | grep -A4 "segment .* rate" >> $OUT; }
IFS=';' read -ra CASES <<< "${SEGMENT_RATE_CASES:-4096 12000;16384 6000;1024 20000;65536 1500;4096 12000 n21;4096 12000 kvh}"
for a in "${CASES[@]}"; do
  run $EXE rate $a stream $DIR
done
run $EXE rate 4096 100 $DIR
PRONTO_SHIM_THREADS=1 $EXE rate 4096 3000 stream $DIR 2>&1 | grep -A4 "segment .* rate" | sed "s/^/PRONTO_SHIM_THREADS=1 /" >> $OUT
cat $OUT
