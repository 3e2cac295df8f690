#!/bin/bash
# tools/tlb_probe.hip, P copies at once, R rounds per variant -> gpurun_out/tlb_probe.txt
root=$(cd "$(dirname "$0")/.." && pwd); cd $root || exit 1
P=${1:-16}; R=${2:-6}; I=${3:-30}
mkdir -p gpurun_out tools/bin
[ -x tools/bin/tlb_probe ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/bin/tlb_probe tools/tlb_probe.hip -lpthread || exit 1
{
for v in "1 0" "0 0" "1 1"; do
  bad=0; runs=0
  for r in $(seq 1 $R); do
    pids=""
    for p in $(seq 1 $P); do timeout 120 tools/bin/tlb_probe $I $v > gpurun_out/.tlb_$p.txt 2>&1 & pids="$pids $!"; done
    for p in $pids; do wait $p || bad=$((bad+1)); runs=$((runs+1)); done
    grep -h -v ": 0 bad" gpurun_out/.tlb_*.txt | head -4
  done
  echo "side thread / allocated once = $v: $bad bad processes of $runs ($P at once, $I iterations each); a good one: $(grep -h ': 0 bad' gpurun_out/.tlb_*.txt | head -1)"
done
} 2>&1 | tee gpurun_out/tlb_probe.txt
rm -f gpurun_out/.tlb_*.txt
