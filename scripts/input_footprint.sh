#!/bin/bash
# 21-state step against the number of DISTINCT sensor blocks it is fed (bench.py streams W + K of them): while state + inputs
# fit the 256 MB memory-side cache everything is cache-resident, beyond it every launch reads its sensor data from HBM.
#   bash scripts/input_footprint.sh > profiles/rNN_n21_input_footprint.txt      (on the GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
f() { python3 bench.py --n-states 21 --no-cpu-baseline --no-cache-busting "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step wall, kernel %.2f us, frac %.3f, %s' % (d['ms_per_step']*1e3, d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['config']['kernel']))"; }
for st in "--steps 10 --warmup 2" "--steps 30 --warmup 5" "--steps 60 --warmup 5" "--steps 200 --warmup 20"; do echo "64k filters, $st: $(f $st)"; done
for b in 32768 49152 98304 131072; do echo "$b filters, --steps 200 --warmup 20: $(f --batch-per-gpu $b)"; done
echo "64k filters, two-wave kernel (PRONTO_BATCH_QUAD21=0), --steps 200 --warmup 20: $(PRONTO_BATCH_QUAD21=0 f)"
