#!/bin/bash
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r3f
mkdir -p $out
cd $root
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -25 $out/tests.log; echo "pytest rc $rc"
for lib in old base; do
  MM_HIP_LIB=$( [ "$lib" = "base" ] && echo "" || echo minimod_amd/lib/var/$lib.so ) timeout 600 python bench.py --config C5 --steps 17 --warmup 2 --reps 3 --no-e2e --no-cpu-baseline --no-extra 2>$out/c5_$lib.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib C5 us/batch %.2f frac %.4f' % (1e3*d['roofline']['kernel_ms_per_batch'], d['roofline']['frac']))"
done
python tools/c5_finalize.py 2>&1 | tail -5
