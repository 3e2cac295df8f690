#!/bin/bash
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r3m
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests/test_hip_stream_gpu.py tests/test_hip_parity.py tests/test_hip_modes_gpu.py tests/test_hip_view_gpu.py tests/test_cli_gpu.py -x -q -k "not full_size" > $out/tests.log 2>&1; echo "pytest rc $?"; tail -3 $out/tests.log
tools/ab.sh 3 "" prev base | tail -2
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o freq -- python3 $root/bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-e2e --no-extra --reps 1 > /dev/null 2>&1
  cp $out/pmc_$c/freq_counter_collection.csv $out/freq_pmc_$c.csv 2>/dev/null; rm -rf $out/pmc_$c
done
python3 $root/tools/pmc_traffic.py $out/freq_pmc_FETCH_SIZE.csv $out/freq_pmc_WRITE_SIZE.csv 16 57154619.75 > $out/traffic_c2.json; python3 -c "
import json; d=json.load(open('$out/traffic_c2.json')); print('traffic MB/batch', d['hbm_bytes_per_batch']/1e6, 'ratio', d['ratio'], d['raw_kb_per_launch']['k_stream_reads'])"
