#!/bin/bash
root=$1
python3 - <<PY
import os, sys
sys.path.insert(0, "$root")
from minimod_amd import synth
ref = synth.reference(3, 400 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=49152, with_order=False) for i in range(12)]
os.makedirs("/tmp/tlb", exist_ok=True)
synth.write_bam_parallel("/tmp/tlb/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/tlb/s.fa", "chrS", ref)
PY
for i in 1 2 3; do
  MM_TIMELINE=1 $root/minimod_amd/bin/minimod freq -b -c "m[CG]" -t 16 --gpu-ingest -o /tmp/tlb/o.bed /tmp/tlb/s.fa /tmp/tlb/s.bam 2>&1 | grep -v Entries | grep "timeline\|Real time\|GPU runtime\|contexts loaded\|Reference genome"
  echo ==
done
rm -rf /tmp/tlb
