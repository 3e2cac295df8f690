#!/bin/bash
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r3h
mkdir -p $out
cd $root
timeout 2400 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; rc=$?
tail -40 $out/tests.log; echo "pytest rc $rc"
