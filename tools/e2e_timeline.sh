#!/bin/bash
# the CLI's timeline (MM_TIMELINE=1) on the 12-Gbases end-to-end input, without and with --gpu-inflate: tools/e2e_timeline.sh <outdir> [gbases]
out=$1; gb=${2:-12}; mkdir -p $out
MM_TIMELINE=1 MM_LOADER_TIMING=1 MM_E2E_STDERR=$out/cli_plain.err timeout 900 python bench.py --e2e-gbases $gb > $out/e2e_plain.json 2> $out/e2e_plain.err
MM_TIMELINE=1 MM_LOADER_TIMING=1 MM_E2E_CLI_FLAGS=--gpu-inflate MM_E2E_STDERR=$out/cli_gpu_inflate.err timeout 900 python bench.py --e2e-gbases $gb > $out/e2e_gpu_inflate.json 2> $out/e2e_gpu_inflate.err
for f in $out/cli_plain.err $out/cli_gpu_inflate.err; do echo "== $f"; grep -v Entries $f | grep "timeline\|Real time\|loader\]\|time:"; grep Entries $f | head -1; done
