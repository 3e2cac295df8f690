#!/bin/bash
# the tied CLI run (-c m[CG],h[CG]: the reference's row order replayed on the device) on a 1.5-Gbase file under the kernel trace: which kernels the replay costs
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/${1:-r5t}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import os, sys
sys.path.insert(0, "$root")
from minimod_amd import synth
ref = synth.reference(3, 48 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=49152, with_order=False) for i in range(12)]
os.makedirs("/tmp/r5tied", exist_ok=True)
synth.write_bam_parallel("/tmp/r5tied/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/r5tied/s.fa", "chrS", ref)
PY
export MM_FULL_TEARDOWN=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_tied -o cli -- $root/minimod_amd/bin/minimod freq -c "m[CG],h[CG]" -m 0.8,0.7 -t 16 --gpu-ingest -o /tmp/r5tied/o.tsv /tmp/r5tied/s.fa /tmp/r5tied/s.bam > $out/cli_tied.log 2>&1
cp $out/ks_tied/cli_kernel_stats.csv $out/cli_tied_kernel_stats.csv 2>/dev/null
rm -rf $out/ks_tied /tmp/r5tied
grep "tie order\|Row order replay\|Real time" $out/cli_tied.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/cli_tied_kernel_stats.csv")))
for r in rows[:22]: print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), ("%.2f ms" % (float(r["TotalDurationNs"]) / 1e6)).rjust(10), r["Percentage"].rjust(6))
PY
