#!/bin/bash
# PC sampling (rocprofv3, beta) of one command: where the wavefronts of its kernels are when the sampler looks.  tools/pcsample.sh <tag> <name> <kernel regex> <command...>
# Output: gpurun_out/<tag>/pcs_<name>_top.txt -- samples per instruction offset of the matching kernels, the hottest first (join with llvm-objdump -d of the code object).
root=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; name=$2; rx=$3; shift; shift; shift
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rm -rf /tmp/pcs_$name
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval ${PCS_INTERVAL:-1000} --kernel-trace --output-format csv -d /tmp/pcs_$name -o pcs -- "$@" > $out/pcs_$name.log 2>&1
echo "rc $?" >> $out/pcs_$name.log
ls -la /tmp/pcs_$name /tmp/pcs_$name/* 2>/dev/null | head -20 >> $out/pcs_$name.log
f=$(ls /tmp/pcs_$name/*pc_sampling*.csv /tmp/pcs_$name/*/*pc_sampling*.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  head -3 "$f" > $out/pcs_${name}_head.txt
  python3 - "$f" "$rx" > $out/pcs_${name}_top.txt <<'PY'
import csv, sys, collections, re
f, rx = sys.argv[1], sys.argv[2]
rows = csv.DictReader(open(f))
cnt = collections.Counter(); total = 0
cols = None
for r in rows:
    if cols is None: cols = list(r.keys())
    total += 1
    key = (r.get("Instruction_Comment") or "", r.get("Instruction") or "", r.get("Code_Object_Id") or "", r.get("Code_Object_Offset") or "")
    cnt[key] += 1
print("columns:", cols)
print("samples:", total)
for k, v in cnt.most_common(400):
    print(v, *k, sep="\t")
PY
  cp "$f" $out/pcs_${name}_samples.csv 2>/dev/null
  gzip -f $out/pcs_${name}_samples.csv
fi
tail -5 $out/pcs_$name.log
