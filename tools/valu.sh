#!/bin/bash
# VALU / SALU wave instructions of k_stream_reads<.., false> in the timed launch, per read
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  out=$root/gpurun_out/valu_$(basename $lib .so); mkdir -p $out
  MM_HIP_LIB=$( [ "$lib" = "base" ] && echo "" || echo $root/minimod_amd/lib/var/$lib.so ) timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $out/p -o k -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra > /dev/null 2>&1
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("$out/p/k_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0]
    if "k_stream_reads" in k and k.rstrip().endswith("false, false, false, false>"): acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])   # (the timed instantiation, not the tally pass's)
best = max(acc.values(), key=lambda d: d.get("SQ_INSTS_VALU", 0))
print("$lib", "VALU/read %.0f SALU/read %.0f; vector unit busy %.0f %% of the SIMDs' time (4 cycles an instruction, %d SIMDs, wave-cycles / waves a SIMD)" % (best["SQ_INSTS_VALU"]/81920, best["SQ_INSTS_SALU"]/81920, 100.0 * best["SQ_INSTS_VALU"] / (best["SQ_WAVE_CYCLES"] / 7.0), 1024))
PY
  rm -rf $out
done
