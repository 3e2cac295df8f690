#!/bin/bash
# the resident set's big mappings at the CLI's marks (MM_TIMELINE=2), on the 1.5-Gbase file with a 400-Mb reference
root=$(cd "$(dirname "$0")/.." && pwd)
python3 - <<PY
import os, sys
sys.path.insert(0, "$root")
from minimod_amd import synth
ref = synth.reference(3, 400 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=98304, with_order=False) for i in range(24)]
os.makedirs("/tmp/tlb", exist_ok=True)
synth.write_bam_parallel("/tmp/tlb/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/tlb/s.fa", "chrS", ref)
PY
MM_TIMELINE=2 $root/minimod_amd/bin/minimod freq -b -c "m[CG]" -t 16 --gpu-ingest -o /tmp/tlb/o.bed /tmp/tlb/s.fa /tmp/tlb/s.bam 2>&1 | grep "timeline\|Real time" | cut -c1-200 > $root/gpurun_out/cli_rss.txt
rm -rf /tmp/tlb
