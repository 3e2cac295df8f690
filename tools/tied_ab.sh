#!/bin/bash
# A/B of two builds of the device library under the tied CLI run (-c m[CG],h[CG]) on a 1.5-Gbase file: tools/tied_ab.sh <other libminimod_hip.so>
# (the CLI finds the library through RUNPATH: LD_LIBRARY_PATH in front of it swaps the build)
root=$(cd "$(dirname "$0")/.." && pwd)
other=$1
mkdir -p /tmp/r5tied /tmp/r5tied/oldlib
cp $other /tmp/r5tied/oldlib/libminimod_hip.so
python3 - <<PY
import os, sys
sys.path.insert(0, "$root")
from minimod_amd import synth
ref = synth.reference(3, 48 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=49152, with_order=False) for i in range(12)]
synth.write_bam_parallel("/tmp/r5tied/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/r5tied/s.fa", "chrS", ref)
PY
cd /tmp
for i in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export LD_LIBRARY_PATH=/tmp/r5tied/oldlib; else unset LD_LIBRARY_PATH; fi
  $root/minimod_amd/bin/minimod freq -c "m[CG],h[CG]" -m 0.8,0.7 -t 16 --gpu-ingest -o /tmp/r5tied/o_$v.tsv /tmp/r5tied/s.fa /tmp/r5tied/s.bam 2> /tmp/r5tied/err_$v.txt
  echo "$v: $(grep -o 'Real time: [0-9.]* sec' /tmp/r5tied/err_$v.txt) | tie order $(grep -o 'sort levels, [0-9.]* ms' /tmp/r5tied/err_$v.txt) | replay $(grep -o 'on the device): [0-9.]* sec' /tmp/r5tied/err_$v.txt) | $(grep -o 'Batch hand-over time: [0-9.]* sec' /tmp/r5tied/err_$v.txt) | $(grep -o 'Data loading time: [0-9.]* sec' /tmp/r5tied/err_$v.txt) | $(grep -o 'Data sorting time: [0-9.]* sec' /tmp/r5tied/err_$v.txt)"
done; done
unset LD_LIBRARY_PATH
cmp /tmp/r5tied/o_new.tsv /tmp/r5tied/o_old.tsv && echo same bytes
rm -rf /tmp/r5tied
