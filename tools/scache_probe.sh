#!/bin/bash
# tools/scache_probe.hip, P copies at once, R rounds a mode: does the scalar data cache hand a kernel what an earlier kernel left at a rewritten address?
#   tools/scache_probe.sh [copies=16] [rounds=6] [iterations=150]   -> gpurun_out/scache_probe.txt
root=$(cd "$(dirname "$0")/.." && pwd); cd $root || exit 1
P=${1:-16}; R=${2:-6}; I=${3:-150}
mkdir -p gpurun_out tools/bin
[ -x tools/bin/scache_probe ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/bin/scache_probe tools/scache_probe.hip || exit 1
{
for mode in 0 1 2; do
  bad=0; runs=0
  for r in $(seq 1 $R); do
    pids=""
    for p in $(seq 1 $P); do timeout 120 tools/bin/scache_probe $mode $I > gpurun_out/.scp_$p.txt 2>&1 & pids="$pids $!"; done
    for p in $pids; do wait $p || bad=$((bad+1)); runs=$((runs+1)); done
    grep -h -v " 0 stale words, 0 wrong totals" gpurun_out/.scp_*.txt | head -4
  done
  echo "mode $mode ($([ $mode = 0 ] && echo s_load || ([ $mode = 1 ] && echo 'agent-scope vector load' || echo 's_dcache_inv + s_load'))): $bad bad processes of $runs ($P at once, $I iterations each)"
done
} 2>&1 | tee gpurun_out/scache_probe.txt
rm -f gpurun_out/.scp_*.txt
