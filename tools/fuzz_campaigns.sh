#!/bin/bash
# the seven random differential campaigns side by side on one GPU: tools/fuzz_campaigns.sh <outdir> <first seed> <scale>
# (scale 1 = 100 mixed / 100 options / 30 long / 60 contigs / 60 shards / 100 errors batches, 80 x 64 inflate streams)
out=$1; s0=${2:-200000}; k=${3:-1}; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 2400 python tools/fuzz_mixed.py   $s0 $((100 * k)) > $out/fuzz_mixed.txt 2>&1 &
timeout 2400 python tools/fuzz_options.py $s0 $((100 * k)) > $out/fuzz_options.txt 2>&1 &
timeout 2400 python tools/fuzz_long.py    $s0 $((30 * k))  > $out/fuzz_long.txt 2>&1 &
timeout 2400 python tools/fuzz_contigs.py $s0 $((60 * k))  > $out/fuzz_contigs.txt 2>&1 &
timeout 2400 python tools/fuzz_shards.py  $s0 $((60 * k))  > $out/fuzz_shards.txt 2>&1 &
timeout 2400 python tools/fuzz_errors.py  $s0 $((100 * k)) > $out/fuzz_errors.txt 2>&1 &
timeout 2400 python tools/fuzz_inflate.py $((s0 / 10)) $((80 * k)) > $out/fuzz_inflate.txt 2>&1 &
wait
for f in $out/fuzz_*.txt; do echo "== $f"; tail -2 $f; done
