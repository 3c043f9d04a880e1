#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/smooth_pmc3; mkdir -p gpurun_out/smooth_pmc3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/smooth_pmc3/a -- python3 scripts/smooth_rate.py > gpurun_out/smooth_pmc3/a.txt 2>&1 || exit 3
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d gpurun_out/smooth_pmc3/b -- python3 scripts/smooth_rate.py > gpurun_out/smooth_pmc3/b.txt 2>&1 || exit 4
echo done
