#!/bin/bash
# round 3, GPU run A: the suite, the driver's line, calibration, phase times, the CLI under rocprofv3
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r3a
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; echo "pytest rc $?" | tee -a $out/gpu_tests.log
tail -5 $out/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/freq_bench.json 2> $out/freq_bench.err; echo "bench rc $?"
python3 -c "
import json
d=json.loads(open('$out/freq_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'frac', d['roofline']['frac'], 'spread', d.get('spread'))
print('coalesce1', d.get('resident_coalesce1')); print('host_path', d.get('host_path')); print('e2e', d.get('end_to_end',{}).get('value'), d.get('end_to_end',{}).get('stages_s'))
"
tools/bin/atomic_calib > $out/atomic_calib.json 2> $out/atomic_calib.err; cat $out/atomic_calib.json
tools/phases.sh "" "--config C3" > $out/phases.txt 2>&1; cat $out/phases.txt
tools/cli_profile.sh $out/cli 25
