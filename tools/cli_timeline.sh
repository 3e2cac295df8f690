#!/bin/bash
# `minimod freq` with MM_TIMELINE=1 on a 48 Mb / 49 152-read synthetic input (the size of the driver's end-to-end leg): where the start-up goes
root=$(cd "$(dirname "$0")/.." && pwd)
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
for i in 1 2 3; do
  MM_TIMELINE=1 $root/minimod_amd/bin/minimod freq -b -c "m[CG]" -t 16 "$@" -o /tmp/r4cli/o.bed /tmp/r4cli/s.fa /tmp/r4cli/s.bam 2>&1 | grep -v Entries | grep "timeline\|Real time\|GPU runtime\|contexts loaded"
  echo ==
done
rm -rf /tmp/r4cli
