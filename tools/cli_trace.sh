#!/bin/bash
# `minimod freq --gpu-ingest` under rocprofv3's kernel and memory-copy trace: tools/cli_trace.sh <tag>
tag=${1:-r4p}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import os, sys
sys.path.insert(0, "$root")
from minimod_amd import synth
ref = synth.reference(3, 48 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=49152, with_order=False) for i in range(12)]
os.makedirs("/tmp/r4cli", exist_ok=True)
synth.write_bam_parallel("/tmp/r4cli/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/r4cli/s.fa", "chrS", ref)
PY
export MM_FULL_TEARDOWN=1   # (the CLI leaves with _exit() otherwise: the profiler would never write its files)
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/ks_cli -o cli -- $root/minimod_amd/bin/minimod freq -b -c "m[CG]" -t 16 --gpu-ingest -o /tmp/r4cli/o.bed /tmp/r4cli/s.fa /tmp/r4cli/s.bam > $out/cli_ingest.log 2>&1
ls $out/ks_cli
cp $out/ks_cli/cli_kernel_stats.csv $out/cli_ingest_kernel_stats.csv 2>/dev/null
cp $out/ks_cli/cli_memory_copy_stats.csv $out/cli_ingest_memory_copy_stats.csv 2>/dev/null
cp $out/ks_cli/cli_memory_copy_trace.csv $out/cli_ingest_memory_copy_trace.csv 2>/dev/null
rm -rf $out/ks_cli /tmp/r4cli
cut -c1-170 $out/cli_ingest_kernel_stats.csv | head -16; cat $out/cli_ingest_memory_copy_stats.csv | cut -c1-170
python3 - <<PY
import csv, collections
try:
    acc = collections.Counter(); n = collections.Counter()
    for r in csv.DictReader(open("$out/cli_ingest_memory_copy_trace.csv")):
        d = r.get("Direction") or r.get("Name") or "?"
        b = int(float(r.get("Bytes", 0) or 0)) if "Bytes" in r else 0
        acc[d] += b; n[d] += 1
    for d in acc: print("%-40s %6d copies %14d bytes" % (d, n[d], acc[d]))
except Exception as e: print("no copy trace:", e)
PY
