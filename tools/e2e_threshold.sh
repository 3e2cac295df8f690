#!/bin/bash
# where the device inflate starts to pay: end-to-end wall seconds by input size, host threads alone against host + device
# usage: tools/e2e_threshold.sh <outdir> <gbases>...
out=$1; shift; mkdir -p $out
for gb in "$@"; do
  for mode in no-gpu-inflate gpu-inflate; do
    MM_E2E_CLI_FLAGS=--$mode timeout 900 python bench.py --e2e-gbases $gb > $out/e2e_${gb}_$mode.json 2> $out/e2e_${gb}_$mode.err
    python - $out/e2e_${gb}_$mode.json $gb $mode <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); g=d["gpu_cli"]
print("%5s Gbases  BAM %.2f GiB  %-16s wall %.3f  load %.3f  identical %s" % (sys.argv[2], d["bam_bytes"]/2**30, sys.argv[3], g["wall_s"], g["stages_s"]["load"], d["parity_vs_cpu"]["byte_identical"]))
PY
  done
done
