#!/bin/bash
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r3i
mkdir -p $out
cd $root
timeout 2400 python -m pytest tests/test_hip_stream_gpu.py tests/test_hip_multicontig_gpu.py -q -k "full_size or devices" > $out/tests.log 2>&1; echo "pytest rc $?"; tail -8 $out/tests.log
timeout 1500 python bench.py --e2e-gbases 12 > $out/e2e_c2_12g.json 2> $out/e2e_c2.err; echo "e2e rc $?"; cat $out/e2e_c2_12g.json | cut -c1-1800
timeout 900 python bench.py --config C3 --e2e-gbases 3 > $out/e2e_c3_3g.json 2> $out/e2e_c3.err; echo "e2e c3 rc $?"; cat $out/e2e_c3_3g.json | cut -c1-2200
tools/phases.sh "" "--config C3" > $out/phases.txt 2>&1; cat $out/phases.txt
timeout 300 python bench.py --mode view --steps 20 --warmup 5 --no-cpu-baseline > $out/view_bench.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$out/view_bench.json').read().strip().splitlines()[-1]); print('view value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])"
