#!/bin/bash
# round 3, GPU run B: parity of the rewritten k_stream_reads, then A/B against the build before it
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r3b
mkdir -p $out
cd $root
timeout 1200 python -m pytest tests/test_hip_stream_gpu.py tests/test_hip_parity.py tests/test_hip_modes_gpu.py tests/test_hip_gather_gpu.py tests/test_hip_skiplists_gpu.py -x -q > $out/tests.log 2>&1; rc=$?
tail -15 $out/tests.log; echo "pytest rc $rc"
if [ $rc -ne 0 ]; then exit 0; fi
MM_DEBUG_OCC=1 timeout 300 python bench.py --steps 20 --warmup 2 --reps 3 --no-e2e --no-cpu-baseline --no-extra 2>&1 | grep -E "workgroups per CU" | head -2
tools/ab.sh 3 "" old base w5 w4 2>&1 | tee $out/ab_c2.txt
tools/ab.sh 2 "--config C3" old base w5 2>&1 | tee $out/ab_c3.txt
tools/ab.sh 2 "--config C5" old base w5 2>&1 | tee $out/ab_c5.txt
