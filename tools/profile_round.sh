#!/bin/bash
# One round's measurements on the GPU box: bench lines and rocprofv3 summaries, written under gpurun_out/<tag>/.
# usage: tools/profile_round.sh <tag>
tag=${1:-r3}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py"
Q="--no-cpu-baseline --no-e2e --no-extra --reps 1"
# the driver's line (with the end-to-end, CPU, host-path and one-launch-per-step legs)
timeout 900 $B --steps 20 --warmup 5 > $out/freq_bench.json 2> $out/freq_bench.err
# per-kernel times of the same steps (no CPU legs, no extra passes, one repetition)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o freq -- $B --steps 20 --warmup 5 $Q > /dev/null 2>&1
cp $out/ks/freq_kernel_stats.csv $out/freq_kernel_stats.csv 2>/dev/null
# HBM traffic: one counter per pass, nothing else traced
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o freq -- $B --steps 16 --warmup 0 $Q > /dev/null 2>&1
  cp $out/pmc_$c/freq_counter_collection.csv $out/freq_pmc_$c.csv 2>/dev/null
done
alg=$(python3 -c "
import json; d=json.loads(open('$out/freq_bench.json').read().strip().splitlines()[-1]); print(d['roofline']['algorithmic_bytes_per_launch'] / d['roofline']['batches_per_launch'])")
python3 $root/tools/pmc_traffic.py $out/freq_pmc_FETCH_SIZE.csv $out/freq_pmc_WRITE_SIZE.csv 16 $alg > $out/traffic_c2.json 2> $out/traffic.err
# the other workloads
for cfg in C3 C5; do
  st=20; [ $cfg = C5 ] && st=17
  timeout 900 $B --config $cfg --steps $st --warmup 5 --no-e2e > $out/${cfg}_bench.json 2> $out/${cfg}_bench.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$cfg -o $cfg -- $B --config $cfg --steps $st --warmup 5 $Q > /dev/null 2>&1
  cp $out/ks_$cfg/${cfg}_kernel_stats.csv $out/${cfg}_kernel_stats.csv 2>/dev/null
done
timeout 300 $B --mode view --steps 20 --warmup 5 > $out/view_bench.json 2> $out/view_bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_view -o view -- $B --mode view --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $out/ks_view/view_kernel_stats.csv $out/view_kernel_stats.csv 2>/dev/null
# the tile pipeline alone, for comparison
timeout 300 $B --steps 20 --warmup 5 --no-stream --no-e2e --no-cpu-baseline --no-extra > $out/freq_bench_no_stream.json 2>/dev/null
rm -rf $out/ks $out/ks_C3 $out/ks_C5 $out/ks_view $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls -la $out
for f in $out/*_bench*.json; do echo "== $f"; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print(d['config']['workload'][:60], '| value %.0f ms/step %.4f frac %.4f launches %s traffic %s' % (d['value'], d['ms_per_step'], r['frac'], r.get('launches'), r.get('traffic')))
    for k in ('spread','resident_coalesce1','host_path'):
        if k in d: print('   ', k, json.dumps(d[k])[:400])
except Exception as e: print('unreadable', e)
"; done
head -8 $out/freq_kernel_stats.csv; head -8 $out/view_kernel_stats.csv; head -6 $out/C5_kernel_stats.csv; cat $out/traffic_c2.json | head -40
