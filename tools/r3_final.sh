#!/bin/bash
# the round's final measurements, all under gpurun_out/r3n/
root=$(pwd); out=$root/gpurun_out/r3n; mkdir -p $out
bash tools/profile_round.sh r3n > $out/profile_round.log 2>&1
bash tools/sq.sh r3n_sq > $out/sq.txt 2>&1
bash tools/cli_profile.sh $out/cli 25 > $out/cli_profile.log 2>&1
MM_LOADER_TIMING=1 timeout 900 python bench.py --e2e-gbases 12 > $out/e2e_c2_12g.json 2> $out/e2e_c2.err
MM_LOADER_TIMING=1 timeout 900 python bench.py --config C3 --e2e-gbases 3 > $out/e2e_c3_3g.json 2> $out/e2e_c3.err
tail -30 $out/profile_round.log
