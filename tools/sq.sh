#!/bin/bash
# SQ counters of one bench configuration: tools/sq.sh <tag> <bench args...>
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT"; do
  tagc=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p -o k -- python3 $root/bench.py --steps 32 --warmup 0 --reps 1 --no-cpu-baseline --no-e2e --no-extra "$@" > /dev/null 2> $out/err_$tagc.txt
  cp $out/p/k_counter_collection.csv $out/cc_$tagc.csv 2>/dev/null
  rm -rf $out/p
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob("$out/cc_*.csv")):
    best = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        key = (k, r["Counter_Name"])
        v = float(r["Counter_Value"])
        # the biggest dispatch of each kernel = the 32-batch launch
        did = r["Dispatch_Id"]
        acc[(k, did)][r["Counter_Name"]] += v
byk = collections.defaultdict(dict)
for (k, did), d in acc.items():
    for c, v in d.items():
        if v > byk[k].get(c, 0): byk[k][c] = v
for k, d in byk.items():
    if "stream" in k or "call_tiles" in k or "scan_reads" in k or "sum_tiles" in k:
        print(k)
        for c, v in sorted(d.items()): print("    %-24s %.3e" % (c, v))
PY
