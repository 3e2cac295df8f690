#!/bin/bash
# quick experiment loop on the GPU box: bench.py variants without the CPU / end-to-end legs, one line each.
# An argument is a string of bench.py options; a leading LIB=<path> selects another build of the device library.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  lib=""
  case "$v" in LIB=*) lib="${v%% *}"; lib="${lib#LIB=}"; v="${v#* }"; [ "$v" = "LIB=$lib" ] && v="";; esac
  MM_HIP_LIB="$lib" timeout 300 python bench.py --steps 50 --warmup 5 --no-e2e --no-cpu-baseline --no-extra $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.0f  ms/step %.4f  kernel_ms/batch %.4f  frac %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms_per_batch'], r['frac']))"
done
