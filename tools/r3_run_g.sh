#!/bin/bash
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r3g
mkdir -p $out
cd $root
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -6 $out/tests.log; echo "pytest rc $rc"
(python tools/fuzz_shards.py 40000 150 2>&1 | tail -3) &
(python tools/fuzz_options.py 40000 100 2>&1 | tail -3) &
(python tools/fuzz_mixed.py 40000 100 2>&1 | tail -3) &
wait
