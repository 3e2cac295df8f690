#!/bin/bash
# rocprofv3 kernel trace of the product CLI on a C2-shape BAM: which kernels the CLI's -K batches reach.
# usage: tools/cli_profile.sh <out dir> [batches of 4096 reads] [extra CLI flags...]
out=$1; nb=${2:-25}; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
case "$out" in /*) ;; *) out=$PWD/$out;; esac
mkdir -p $out /tmp/clip
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import sys, os
sys.path.insert(0, "$root")
from minimod_amd import synth
ref = synth.reference(3, 48 << 20)
nb = $nb
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=4096 * nb, with_order=False) for i in range(nb)]
synth.write_bam_parallel("/tmp/clip/s.bam", [("chrS", len(ref))], bs, threads=32)
synth.write_fasta("/tmp/clip/s.fa", "chrS", ref)
print("bases", sum(b["n_bases"] for b in bs))
PY
timeout 600 rocprofv3 --kernel-trace $MM_PROF_EXTRA --stats --output-format csv -d $out/p -o cli -- $root/minimod_amd/bin/minimod freq -b -c 'm[CG]' -m 0.8 -K 4096 -B 200M -t 64 "$@" -o /tmp/clip/out.bed /tmp/clip/s.fa /tmp/clip/s.bam > $out/cli_stdout.txt 2> $out/cli_stderr.txt
cp $out/p/cli_kernel_stats.csv $out/cli_kernel_stats.csv 2>/dev/null
for f in $out/p/cli_memory_copy_stats.csv $out/p/cli_memory_copy_trace.csv; do [ -f $f ] && cp $f $out/; done
rm -rf $out/p
grep -E "GPU launches|time:" $out/cli_stderr.txt
head -12 $out/cli_kernel_stats.csv
