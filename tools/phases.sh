#!/bin/bash
# where a wavefront of k_stream_reads spends its time (s_memrealtime laps of a -DMM_STREAM_TIMING build, minimod_amd/lib/var/hip_kf_timing.so:
#   MM_HIP_LIB=$PWD/minimod_amd/lib/var/hip_kf_timing.so MM_HIP_DEFS=-DMM_STREAM_TIMING python -c 'from minimod_amd import build; build.build_hip()'):
# tools/phases.sh "" "--config C3" ...
cd "$(dirname "$0")/.."
for cfg in "$@"; do
MM_HIP_LIB=minimod_amd/lib/var/hip_kf_timing.so timeout 300 python bench.py --steps 32 --warmup 0 --no-e2e --no-cpu-baseline --no-extra $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ph=d.get('phase_cycles_timed',[0]*12)[3:]
names=['record+cigar','headers','parse','dirwin','locate','cigwin','finish','upkeep','groupsetup']
tot=sum(ph) or 1
print('$cfg', d['config'].get('routing', ''))
for n,v in zip(names,ph): print('   %-14s %6.1f%%' % (n, 100.0*v/tot))
"
done
