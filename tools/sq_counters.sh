#!/bin/bash
# SQ instruction / cycle counters of the hot-path kernels (mean per dispatch of the timed launch), for tools/profile_round.sh
tag=${1:-sq}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-e2e --no-extra $2"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o sq -- $B > /dev/null 2>&1
  cp $out/p$i/sq_counter_collection.csv $out/sq_pass$i.csv 2>/dev/null
  rm -rf $out/p$i
done
python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("$out/sq_pass*.csv")):
    last = {}
    for r in csv.DictReader(open(f)):
        for k in ("k_plan_items", "k_scan_reads", "k_sum_tiles", "k_call_tiles"):
            if k in r["Kernel_Name"]:
                last[(k, r["Counter_Name"])] = float(r["Counter_Value"])     # the last dispatch = the timed launch
    agg.update(last)
names = sorted({c for _, c in agg})
with open("$out/sq_counters.csv", "w") as o:
    o.write("kernel," + ",".join(names) + "\n")
    for k in ("k_plan_items", "k_scan_reads", "k_sum_tiles", "k_call_tiles"):
        o.write(k + "," + ",".join("%.0f" % agg.get((k, c), float("nan")) for c in names) + "\n")
print(open("$out/sq_counters.csv").read())
PY
