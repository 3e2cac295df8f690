#!/bin/bash
# Round 5, second pass: what changed after tools/r5_final.sh ran (the inflate's new symbol loop, view / -c '*' through the device reader) -- the
# inflate alone (three launch sizes, SQ counters of round 4's loop and of this one), the CLI under the kernel trace, the driver's line
# (its end-to-end leg), the 12-Gbase job and the tied C3 job end to end.   usage: tools/r5_final2.sh <tag>
tag=${1:-r5y}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
for n in 6144 4096 2048; do python3 tools/inflate_bench.py 24576 $n 2>&1 | tail -1; done > $out/inflate_bench.txt
[ -f minimod_amd/lib/var/v1.so ] && for n in 6144 4096 2048; do MM_HIP_LIB=$root/minimod_amd/lib/var/v1.so python3 tools/inflate_bench.py 24576 $n 2>&1 | tail -1; done > $out/inflate_bench_round4_loop.txt
bash tools/inflate_sq.sh $tag/sq_new > /dev/null 2>&1; cp $out/sq_new/inflate_sq_counters.txt $out/inflate_sq_counters.txt
[ -f minimod_amd/lib/var/v1.so ] && { bash tools/inflate_sq.sh $tag/sq_v1 MM_HIP_LIB=$root/minimod_amd/lib/var/v1.so > /dev/null 2>&1; cp $out/sq_v1/inflate_sq_counters.txt $out/inflate_sq_counters_round4_loop.txt; }
rm -rf $out/sq_new $out/sq_v1
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py"
python3 - <<PY
import os, sys
sys.path.insert(0, "$root")
from minimod_amd import synth
ref = synth.reference(3, 48 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=49152, with_order=False) for i in range(12)]
os.makedirs("/tmp/r5cli", exist_ok=True)
synth.write_bam_parallel("/tmp/r5cli/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/r5cli/s.fa", "chrS", ref)
PY
export MM_FULL_TEARDOWN=1   # (the CLI leaves with _exit() otherwise: the profiler would never write its files)
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/ks_cli -o cli -- $root/minimod_amd/bin/minimod freq -b -c "m[CG]" -t 16 --gpu-ingest -o /tmp/r5cli/o.bed /tmp/r5cli/s.fa /tmp/r5cli/s.bam > $out/cli_ingest.log 2>&1
cp $out/ks_cli/cli_kernel_stats.csv $out/cli_ingest_kernel_stats.csv 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_view -o cli -- $root/minimod_amd/bin/minimod view -c "m[CG]" -t 16 --gpu-ingest -o /tmp/r5cli/o.tsv /tmp/r5cli/s.fa /tmp/r5cli/s.bam > $out/cli_view_ingest.log 2>&1
cp $out/ks_view/cli_kernel_stats.csv $out/cli_view_ingest_kernel_stats.csv 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_wild -o cli -- $root/minimod_amd/bin/minimod freq -c "*" -t 16 --gpu-ingest -o /tmp/r5cli/o2.tsv /tmp/r5cli/s.fa /tmp/r5cli/s.bam > $out/cli_wildcard_ingest.log 2>&1
cp $out/ks_wild/cli_kernel_stats.csv $out/cli_wildcard_ingest_kernel_stats.csv 2>/dev/null
unset MM_FULL_TEARDOWN
rm -rf $out/ks_cli $out/ks_view $out/ks_wild /tmp/r5cli
( time timeout 900 $B --steps 20 --warmup 5 > $out/freq_bench.json 2> $out/freq_bench.err ) 2> $out/freq_bench.time
MM_E2E_STDERR=$out/e2e_c2_12g_cli_log.txt timeout 900 $B --e2e-gbases 12 > $out/e2e_c2_12g.json 2> $out/e2e_c2_12g.err
MM_E2E_STDERR=$out/e2e_c3_3g_cli_log.txt timeout 900 $B --config C3 --e2e-gbases 3 > $out/e2e_c3_3g.json 2> $out/e2e_c3_3g.err
python3 -c "
import json
d=json.loads(open('$out/e2e_c2_12g.json').read().strip().splitlines()[-1]); g=d['gpu_cli']; print('12G wall', g['wall_s'], g['stages_s'], 'cpu', d['cpu_port']['wall_s'], d['parity_vs_cpu']['byte_identical'])
d=json.loads(open('$out/e2e_c3_3g.json').read().strip().splitlines()[-1]); g=d['gpu_cli']; r=d['reference_order_replay']; print('C3 3G canonical', g['wall_s'], 'tied default', r['wall_s'], r['replay_s'], r['replay_on'], 'host replay', r['host_replay'])
d=json.loads(open('$out/freq_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('headline value %.0f ms/step %.4f frac %.4f' % (d['value'], d['ms_per_step'], r['frac'])); print('  e2e', d['end_to_end']['wall_s'], d['end_to_end']['stages_s'], 'cpu', d['cpu_baseline_e2e']['t_all']['wall_s'])
"
cat $out/inflate_bench.txt $out/inflate_bench_round4_loop.txt; grep "INSTS" $out/inflate_sq_counters.txt $out/inflate_sq_counters_round4_loop.txt
head -5 $out/cli_ingest_kernel_stats.csv | cut -c1-150; head -6 $out/cli_view_ingest_kernel_stats.csv | cut -c1-150; grep "gpu-ingest\]\|Real time" $out/cli_view_ingest.log $out/cli_wildcard_ingest.log | cut -c1-250
