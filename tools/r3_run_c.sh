#!/bin/bash
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r3c
mkdir -p $out
cd $root
tools/ab.sh 2 "" base noatomic noref noboth 2>&1 | tee $out/ab_abl_c2.txt
tools/ab.sh 2 "--config C3" base noatomic noref noboth 2>&1 | tee $out/ab_abl_c3.txt
tools/valu.sh old base 2>&1 | tee $out/valu.txt
tools/phases.sh "" "--config C3" > $out/phases.txt 2>&1; cat $out/phases.txt
for lib in old base; do
  MM_HIP_LIB=$( [ "$lib" = "base" ] && echo "" || echo minimod_amd/lib/var/$lib.so ) timeout 600 python bench.py --config C5 --steps 17 --warmup 2 --reps 3 --no-e2e --no-cpu-baseline --no-extra 2>$out/c5_$lib.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib C5 us/batch %.2f frac %.4f' % (1e3*d['roofline']['kernel_ms_per_batch'], d['roofline']['frac']), d['config']['routing'])"
done
tail -3 $out/c5_base.err
