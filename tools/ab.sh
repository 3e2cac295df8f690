#!/bin/bash
# A/B of device library builds: tools/ab.sh <reps> <bench args or ""> lib1 lib2 ...   ("base" = the in-tree build)
cd "$(dirname "$0")/.." || exit 1
reps=$1; shift; bargs=$1; shift
for r in $(seq $reps); do
  for lib in "$@"; do
    path=""; [ "$lib" != "base" ] && path=minimod_amd/lib/var/$lib.so
    MM_HIP_LIB="$path" timeout 300 python bench.py --steps 50 --warmup 5 --no-e2e --no-cpu-baseline --no-extra $bargs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '%.2f' % (1e3*d['roofline']['kernel_ms_per_batch']))" >> /tmp/ab_$$.txt
  done
done
python - <<PY
import collections, statistics
acc = collections.defaultdict(list)
for l in open("/tmp/ab_$$.txt"):
    k, v = l.split(); acc[k].append(float(v))
for k, v in acc.items(): print("%-16s median %.2f  min %.2f  max %.2f  (%d runs)  us/batch  %s" % (k, statistics.median(v), min(v), max(v), len(v), "$bargs"))
PY
rm -f /tmp/ab_$$.txt
