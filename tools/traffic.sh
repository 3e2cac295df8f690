#!/bin/bash
# HBM traffic of a bench configuration's timed launch: tools/traffic.sh <outdir tag> <name> <steps> <bench args...>
#   -> gpurun_out/<tag>/traffic_<name>.json  (copy to profiles/ to give the bench line its roofline.traffic)
tag=$1; name=$2; steps=$3; shift 3
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py"
Q="--no-cpu-baseline --no-e2e --no-extra --reps 1"
timeout 300 $B --steps $steps --warmup 0 $Q "$@" > $out/bench_$name.json 2> $out/bench_$name.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${name}_$c -o t -- $B --steps $steps --warmup 0 $Q "$@" > /dev/null 2>&1
  cp $out/pmc_${name}_$c/t_counter_collection.csv $out/${name}_pmc_$c.csv 2>/dev/null
  rm -rf $out/pmc_${name}_$c
done
alg=$(python3 -c "
import json; d=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print(r['algorithmic_bytes_per_launch'] / r.get('batches_per_launch', 1.0))")
python3 $root/tools/pmc_traffic.py $out/${name}_pmc_FETCH_SIZE.csv $out/${name}_pmc_WRITE_SIZE.csv $steps $alg "$name: bench.py $*" > $out/traffic_$name.json 2> $out/traffic_$name.err
python3 -c "
import json; d=json.load(open('$out/traffic_$name.json')); print('$name', 'bytes/batch %.1f MB' % (d['hbm_bytes_per_batch']/1e6), 'algorithmic %.1f MB' % (d['algorithmic_bytes_per_batch']/1e6), 'ratio %.2f' % d['ratio'])"
