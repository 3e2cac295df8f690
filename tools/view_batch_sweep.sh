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
for rep in 1 2; do
for cfg in "512 16000000" "1024 32000000" "2048 64000000" "2048 150000000" "host"; do
  set -- $cfg
  if [ $1 = host ]; then
    $B view -c "m[CG]" -t 16 --no-gpu-ingest -o /tmp/bc/a.out /tmp/bc/s.fa /tmp/bc/s.bam 2> /tmp/bc/a.err
  else
    MM_INGEST_MAX_BLOCKS=$1 MM_INGEST_TARGET_BASES=$2 $B view -c "m[CG]" -t 16 --gpu-ingest -o /tmp/bc/a.out /tmp/bc/s.fa /tmp/bc/s.bam 2> /tmp/bc/a.err
  fi
  echo "$cfg: $(grep -o 'Real time: [0-9.]* sec' /tmp/bc/a.err) | $(grep -o 'Data loading time: [0-9.]* sec\|Data output time: [0-9.]* sec\|Data processing time: [0-9.]* sec' /tmp/bc/a.err | tr '\n' ' ')"
done; done
rm -rf /tmp/bc
