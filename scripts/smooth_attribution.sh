#!/bin/bash
# Where the RTS smoother step's time goes (smoother): builds pb_smooth.hip with parts of k_smooth_reg compiled
# out (rbis_smooth.hpp, SM_* flags: results are garbage, only the time means something), links each variant against the
# other objects of the library and times scripts/smooth_rate.py with it (PRONTO_BATCH_LIB).  Also the occupancy probe:
# PRONTO_SMOOTH_LDS_PAD asks for 30 KB more LDS per workgroup, so only ONE workgroup fits a CU instead of two.
#   on the build host:  bash scripts/smooth_attribution.sh build
#   on the GPU box:     bash scripts/smooth_attribution.sh run > gpurun_out/smooth_attribution.txt
set -u
cd "$(dirname "$0")/.."
V="full:  fact:-DSM_SKIP_FACT subst:-DSM_SKIP_SUBST prod:-DSM_SKIP_PROD quat:-DSM_SKIP_QUAT \
   compute:-DSM_SKIP_FACT,-DSM_SKIP_SUBST,-DSM_SKIP_PROD,-DSM_SKIP_QUAT staging_only:-DSM_COPY_ONLY noload:-DSM_NO_LOAD nomem:-DSM_NO_LOAD,-DSM_NO_STORE empty:-DSM_EMPTY occ3:-DSM_OCC3 \
   skew2:-DSM_SKEW=2 skew4:-DSM_SKEW=4 skew6:-DSM_SKEW=6"
V=${SM_VARIANTS:-$V}   # e.g. SM_VARIANTS="full: skew4:-DSM_SKEW=4"
D=gpurun_scratch/smooth_attr
if [ "${1:-}" = build ]; then
  mkdir -p $D
  O=pronto_amd/lib/obj
  for v in $V; do
    n=${v%%:*}; fl=$(echo ${v#*:} | tr ',' ' ')
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DPB_EXPERIMENTS -Ipronto_amd/csrc $fl -c -o $D/pb_smooth_$n.o pronto_amd/csrc/pb_smooth.hip || exit 1
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DPB_EXPERIMENTS -mllvm -disable-machine-licm -Ipronto_amd/csrc $fl -c -o $D/pb_smooth_wide_$n.o pronto_amd/csrc/pb_smooth_wide.hip || exit 1
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o $D/lib_$n.so $O/pronto_batch.o $O/pb_step.o $O/pb_step_leg15.o $O/pb_step_leg21.o $O/pb_update15.o \
      $O/pb_update21_0.o $O/pb_update21_1.o $O/pb_update21_2.o $O/pb_update_ct.o $D/pb_smooth_$n.o $D/pb_smooth_wide_$n.o || exit 1
    rm -f $D/pb_smooth_$n.o $D/pb_smooth_wide_$n.o
  done
  ls $D
  exit 0
fi
for v in $V; do
  n=${v%%:*}
  echo "== compiled out: $n (${v#*:})"
  PRONTO_BATCH_LIB=$PWD/$D/lib_$n.so python3 scripts/smooth_rate.py 2>&1 | grep smoother
done
echo "== full kernel, one workgroup per CU (PRONTO_SMOOTH_LDS_PAD=30000)"
PRONTO_BATCH_LIB=$PWD/$D/lib_full.so PRONTO_SMOOTH_LDS_PAD=30000 python3 scripts/smooth_rate.py 2>&1 | grep smoother
