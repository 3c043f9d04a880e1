#!/bin/bash
# A/B of the pair kernels at 64k filters: every variant (a list of VAR=value settings, "-" = none) is run REPS times, interleaved,
# and the minimum per row is printed -- single runs on this pool scatter by 1-3 us.   bash scripts/leg_ab.sh 3 - PRONTO_BATCH_LEGPLAN=0
REPS=$1; shift
OUT=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/r04/leg_ab
mkdir -p $OUT; rm -f $OUT/*.txt
for r in $(seq $REPS); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    if [ "$v" = "-" ]; then python3 scripts/leg_rates.py 65536 pairs >> $OUT/v$i.txt 2>/dev/null
    else env $v python3 scripts/leg_rates.py 65536 pairs >> $OUT/v$i.txt 2>/dev/null; fi
  done
done
i=0
for v in "$@"; do
  i=$((i+1)); echo "== variant $i: $v  (min of $REPS)"
  python3 - $OUT/v$i.txt <<'PY'
import sys,collections
best=collections.OrderedDict()
for l in open(sys.argv[1]):
    if " us " not in l: continue
    key=l[:l.index(" us")].rsplit(None,1)[0]; us=float(l[:l.index(" us")].rsplit(None,1)[1])
    best[key]=min(best.get(key,1e9),us)
for k,v in best.items(): print("%s %7.1f us"%(k,v))
PY
done
