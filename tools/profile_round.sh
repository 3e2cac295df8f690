#!/bin/bash
# One round's measurements on the GPU box: bench lines and rocprofv3 summaries, written under gpurun_out/<tag>/.
# usage: tools/profile_round.sh <tag>
tag=${1:-r2}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py"
# the driver's line (with the end-to-end and CPU legs)
timeout 600 $B --steps 20 --warmup 5 > $out/freq_bench.json 2> $out/freq_bench.err
# per-kernel times of the same steps (no CPU legs, no extra passes)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o freq -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra > /dev/null 2>&1
cp $out/ks/freq_kernel_stats.csv $out/freq_kernel_stats.csv 2>/dev/null
# HBM traffic: one counter per pass, nothing else traced
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o freq -- $B --steps 16 --warmup 0 --no-cpu-baseline --no-e2e --no-extra > /dev/null 2>&1
  cp $out/pmc_$c/freq_counter_collection.csv $out/freq_pmc_$c.csv 2>/dev/null
done
# the other workloads
for cfg in C3 C5; do
  timeout 600 $B --config $cfg --steps 20 --warmup 5 --no-e2e > $out/${cfg}_bench.json 2> $out/${cfg}_bench.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$cfg -o $cfg -- $B --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra > /dev/null 2>&1
  cp $out/ks_$cfg/${cfg}_kernel_stats.csv $out/${cfg}_kernel_stats.csv 2>/dev/null
done
timeout 300 $B --mode view --steps 20 --warmup 5 > $out/view_bench.json 2> $out/view_bench.err
# uncoalesced single launches, for comparison with round 1
timeout 300 $B --steps 20 --warmup 5 --coalesce 1 --no-e2e --no-cpu-baseline > $out/freq_bench_coalesce1.json 2>/dev/null
rm -rf $out/ks $out/ks_C3 $out/ks_C5 $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
# what FETCH_SIZE / WRITE_SIZE mean for the kernels' access shapes (tools/fetch_calib.hip)
if [ -x $root/tools/bin/fetch_calib ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/cal_$c -o calib -- $root/tools/bin/fetch_calib > $out/fetch_calib_asked.csv 2>/dev/null
    cp $out/cal_$c/calib_counter_collection.csv $out/fetch_calib_$c.csv 2>/dev/null
    rm -rf $out/cal_$c
  done
fi
# the tile pipeline alone, for comparison
timeout 300 $B --steps 20 --warmup 5 --no-stream --no-e2e --no-cpu-baseline > $out/freq_bench_no_stream.json 2>/dev/null
timeout 300 $B --config C3 --steps 20 --warmup 5 --no-stream --no-e2e --no-cpu-baseline > $out/C3_bench_no_stream.json 2>/dev/null
ls -la $out
for f in $out/*_bench*.json; do echo "== $f"; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print(d['config']['workload'][:60], '| value %.0f ms/step %.4f frac %.4f launches %s' % (d['value'], d['ms_per_step'], r['frac'], r.get('launches')))
except Exception as e: print('unreadable', e)
"; done
head -8 $out/freq_kernel_stats.csv
