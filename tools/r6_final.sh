#!/bin/bash
# Round 6's measurements on the GPU box, in the order they depend on each other: the HBM traffic of every workload's timed launch
# first (PMC passes; the files go to profiles/ so that the bench lines behind them carry roofline.traffic), then the driver's
# line (with config_fracs, the end-to-end legs, the CPU legs), per-kernel times, the inflate kernel alone.
# usage: tools/r6_final.sh <tag>
tag=${1:-r6z}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 300 tools/traffic.sh $tag c2 16
timeout 300 tools/traffic.sh $tag c3 16 --config C3
timeout 300 tools/traffic.sh $tag c5 16 --config C5
timeout 300 tools/traffic.sh $tag view 16 --mode view
timeout 400 tools/traffic.sh $tag c4 16 --config C4
cp $out/traffic_c2.json $out/traffic_c3.json $out/traffic_c5.json $out/traffic_view.json $out/traffic_c4.json $root/profiles/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py"
Q="--no-cpu-baseline --no-e2e --no-extra --reps 1"
( time timeout 900 $B --steps 20 --warmup 5 > $out/freq_bench.json 2> $out/freq_bench.err ) 2> $out/freq_bench.time
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o freq -- $B --steps 20 --warmup 5 $Q > /dev/null 2>&1
cp $out/ks/freq_kernel_stats.csv $out/freq_kernel_stats.csv 2>/dev/null
for cfg in C3 C5; do
  st=20; [ $cfg = C5 ] && st=17
  timeout 900 $B --config $cfg --steps $st --warmup 5 --no-e2e > $out/${cfg}_bench.json 2> $out/${cfg}_bench.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$cfg -o $cfg -- $B --config $cfg --steps $st --warmup 5 $Q > /dev/null 2>&1
  cp $out/ks_$cfg/${cfg}_kernel_stats.csv $out/${cfg}_kernel_stats.csv 2>/dev/null
done
timeout 300 $B --mode view --steps 20 --warmup 5 > $out/view_bench.json 2> $out/view_bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_view -o view -- $B --mode view --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $out/ks_view/view_kernel_stats.csv $out/view_kernel_stats.csv 2>/dev/null
# the CLI with the device loader under the kernel trace (a 1.5-Gbase file: every kernel of the ingestion and of the freq path by name)
python3 - <<PY
import os, sys, subprocess
sys.path.insert(0, "$root")
from minimod_amd import synth
ref = synth.reference(3, 48 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=49152, with_order=False) for i in range(12)]
os.makedirs("/tmp/r6cli", exist_ok=True)
synth.write_bam_parallel("/tmp/r6cli/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/r6cli/s.fa", "chrS", ref)
PY
export MM_FULL_TEARDOWN=1   # (the CLI leaves with _exit() otherwise: the profiler would never write its files)
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/ks_cli -o cli -- $root/minimod_amd/bin/minimod freq -b -c "m[CG]" -t 16 --gpu-ingest -o /tmp/r6cli/o.bed /tmp/r6cli/s.fa /tmp/r6cli/s.bam > $out/cli_ingest.log 2>&1
cp $out/ks_cli/cli_kernel_stats.csv $out/cli_ingest_kernel_stats.csv 2>/dev/null
cp $out/ks_cli/cli_memory_copy_stats.csv $out/cli_ingest_memory_copy_stats.csv 2>/dev/null
unset MM_FULL_TEARDOWN
rm -rf $out/ks $out/ks_C3 $out/ks_C5 $out/ks_view $out/ks_cli /tmp/r6cli
cd $root
python3 tools/inflate_bench.py 24576 6144 > $out/inflate_bench.txt 2>&1
python3 tools/inflate_bench.py 24576 4096 >> $out/inflate_bench.txt 2>&1
python3 tools/inflate_bench.py 8192 2048 >> $out/inflate_bench.txt 2>&1
timeout 400 $B --config C4 --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-extra > $out/C4_bench_n1.json 2> $out/C4_bench_n1.err
timeout 600 tools/sq_dispatch.sh $tag c2 > /dev/null 2>&1
MM_E2E_STDERR=$out/e2e_c2_12g_cli_log.txt timeout 900 $B --e2e-gbases 12 > $out/e2e_c2_12g.json 2> $out/e2e_c2_12g.err
MM_E2E_STDERR=$out/e2e_c3_3g_cli_log.txt timeout 900 $B --config C3 --e2e-gbases 3 > $out/e2e_c3_3g.json 2> $out/e2e_c3_3g.err
python3 -c "
import json
d=json.loads(open('$out/e2e_c2_12g.json').read().strip().splitlines()[-1]); g=d['gpu_cli']; print('12G wall', g['wall_s'], g['stages_s'], 'cpu', d['cpu_port']['wall_s'], d['parity_vs_cpu']['byte_identical'])
d=json.loads(open('$out/e2e_c3_3g.json').read().strip().splitlines()[-1]); g=d['gpu_cli']; r=d['reference_order_replay']; print('C3 3G canonical', g['wall_s'], 'tied default', r['wall_s'], r['replay_s'], r['replay_on'], 'host replay', r['host_replay'])
"
cat $out/sq_dispatch_c2.txt 2>/dev/null | tail -12
for f in $out/*_bench*.json; do echo "== $f"; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print(d['config']['workload'][:60], '| value %.0f ms/step %.4f frac %.4f traffic %s' % (d['value'], d['ms_per_step'], r['frac'], r.get('traffic')))
    for k in ('config_fracs',):
        if k in d: print('   ', k, json.dumps(d[k])[:900])
    if 'end_to_end' in d: print('    e2e', d['end_to_end']['wall_s'], d['end_to_end']['stages_s'], 'cpu', d['cpu_baseline_e2e']['t_all']['wall_s'])
except Exception as e: print('unreadable', e)
"; done
cat $out/freq_bench.time $out/inflate_bench.txt; head -6 $out/cli_ingest_kernel_stats.csv | cut -c1-150; cat $out/cli_ingest_memory_copy_stats.csv 2>/dev/null | cut -c1-150
