#!/bin/bash
# A/B of bench.py flag sets on the in-tree build: tools/ab_flags.sh <reps> "<common bench args>" "<flags A>" "<flags B>" ...
cd "$(dirname "$0")/.." || exit 1
reps=$1; shift; common=$1; shift
for r in $(seq $reps); do
  i=0
  for fl in "$@"; do
    i=$((i+1))
    timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-extra $common $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v$i', '%.2f' % (1e3*d['roofline']['kernel_ms_per_batch']), '%.4f' % d['roofline']['frac'])" >> /tmp/abf_$$.txt
  done
done
python - "$@" <<PY
import collections, statistics, sys
acc = collections.defaultdict(list); fr = collections.defaultdict(list)
for l in open("/tmp/abf_$$.txt"):
    k, v, f = l.split(); acc[k].append(float(v)); fr[k].append(float(f))
for i, fl in enumerate(sys.argv[1:]):
    k = "v%d" % (i + 1); v = acc[k]
    print("%-32s median %.2f  min %.2f  max %.2f us/batch  frac %.3f  (%d runs)  $common" % (fl or "(default)", statistics.median(v), min(v), max(v), statistics.median(fr[k]), len(v)))
PY
rm -f /tmp/abf_$$.txt
