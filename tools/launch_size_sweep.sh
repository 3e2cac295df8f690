#!/bin/bash
# How many reads must a launch hold before k_stream_reads pays?  C2's resident read set, launches of N consecutive -K windows (mm_freq_opts_t.coalesce):
# the roofline fraction of the timed launches against reads a launch -> gpurun_out/r6_launch_size_sweep.txt   
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
echo "C2 (100 000 ONT-shape reads, 15 kb, resident in HBM): roofline fraction of the freq hot path against the reads a launch holds"
echo "  -K     windows a launch   reads a launch   kernel us a batch   Gbases/s   frac   reads done by k_stream_reads"
for K in 4096 512; do
  for N in 1 2 3 4 6 8 12 16 24 32; do
    [ $K = 512 ] && [ $N -lt 4 ] && continue
    timeout 300 python bench.py --batch $K --coalesce $N --steps $(( 4096 / K * 24 )) --warmup 4 --no-e2e --no-cpu-baseline --no-extra --no-host-path --no-config-fracs --reps 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']['routing']
print('  %-6d %-18d %-16d %-19.2f %-10.0f %-6.4f %d of %d' % ($K, $N, $K*$N, 1e3*r['kernel_ms_per_batch'], d['value']/1e3, r['frac'], c['reads_done_by_k_stream_reads'], c['reads']))"
  done
done
} | tee gpurun_out/r6_launch_size_sweep.txt
