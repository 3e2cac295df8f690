#!/bin/bash
# any CLI run on the 1.5-Gbase synthetic file under the kernel trace: tools/cli_trace_any.sh <tag> <name> <minimod arguments in front of the files...>
root=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; name=$2; shift; shift
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ ! -f /tmp/r5any/s.bam ]; then
python3 - <<PY
import os, sys
sys.path.insert(0, "$root")
from minimod_amd import synth
ref = synth.reference(3, 48 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=49152, with_order=False, hp_tags=True, long_insertions=True) if False else synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=49152, with_order=False) for i in range(12)]
os.makedirs("/tmp/r5any", exist_ok=True)
synth.write_bam_parallel("/tmp/r5any/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/r5any/s.fa", "chrS", ref)
PY
fi
export MM_FULL_TEARDOWN=1
set -f
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$name -o cli -- $root/minimod_amd/bin/minimod "$@" -t 16 -o /tmp/r5any/o.out /tmp/r5any/s.fa /tmp/r5any/s.bam > $out/cli_$name.log 2>&1
set +f
cp $out/ks_$name/cli_kernel_stats.csv $out/cli_${name}_kernel_stats.csv 2>/dev/null
rm -rf $out/ks_$name
grep "Real time\|Data sorting\|Data output\|Data loading" $out/cli_$name.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/cli_${name}_kernel_stats.csv")))
for r in rows[:14]: print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), ("%.2f ms" % (float(r["TotalDurationNs"]) / 1e6)).rjust(10), r["Percentage"].rjust(6))
PY
