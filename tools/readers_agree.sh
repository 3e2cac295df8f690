#!/bin/bash
# the device-side reader against the host threads on a 1.5-Gbase file: view, view -c '*', freq -c '*', a tied freq run -- the same bytes?
# usage: tools/readers_agree.sh <repo root>
root=$1
python3 - <<PY
import os, sys
sys.path.insert(0, "$root")
from minimod_amd import synth
ref = synth.reference(3, 48 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=49152, with_order=False) for i in range(12)]
os.makedirs("/tmp/bc", exist_ok=True)
synth.write_bam_parallel("/tmp/bc/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/bc/s.fa", "chrS", ref)
PY
B=$root/minimod_amd/bin/minimod
for mode in "view -c m[CG]" "view -c *" "freq -c *" "freq -c m[CG],h[CG] --insertions"; do
  set -f
  $B $mode -t 16 --gpu-ingest -o /tmp/bc/a.out /tmp/bc/s.fa /tmp/bc/s.bam 2> /tmp/bc/a.err; ta=$(grep -o "Real time: [0-9.]* sec" /tmp/bc/a.err)
  $B $mode -t 16 --no-gpu-ingest -o /tmp/bc/b.out /tmp/bc/s.fa /tmp/bc/s.bam 2> /tmp/bc/b.err; tb=$(grep -o "Real time: [0-9.]* sec" /tmp/bc/b.err)
  set +f
  echo "$mode: device reader $ta ($(grep -c 'gpu-ingest\]' /tmp/bc/a.err) marks), host reader $tb; $(wc -c < /tmp/bc/a.out) bytes; identical: $(cmp -s /tmp/bc/a.out /tmp/bc/b.out && echo yes || echo NO)"
done
rm -rf /tmp/bc
