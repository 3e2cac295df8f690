#!/bin/bash
# SQ counters of k_bgzf_inflate on a C2-shape BAM's blocks (tools/inflate_bench.py): tools/inflate_sq.sh <tag> [MM_HIP_LIB=...]
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do export "$v"; done
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT"; do
  tagc=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p -o k -- python3 $root/tools/inflate_bench.py 4096 1536 > $out/bench_$tagc.txt 2> $out/err_$tagc.txt
  cp $out/p/k_counter_collection.csv $out/cc_$tagc.csv 2>/dev/null
  rm -rf $out/p
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob("$out/cc_*.csv")):
    for r in csv.DictReader(open(f)):
        if "k_bgzf_inflate" not in r["Kernel_Name"]: continue
        acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
# mean over the dispatches of every counter
tot = collections.defaultdict(list)
for did, d in acc.items():
    for c, v in d.items(): tot[c].append(v)
with open("$out/inflate_sq_counters.txt", "w") as o:
    for c in sorted(tot):
        line = "%-24s %14.0f per launch of 1536 blocks (%d launches)" % (c, sum(tot[c]) / len(tot[c]), len(tot[c]))
        print(line); o.write(line + "\n")
PY
