cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_hip_modes_gpu.py -x -q -m gpu 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp -o e -- python3 bench.py --no-cpu-baseline --no-extra --steps 30 > gpurun_out/exp.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/exp/e_kernel_stats.csv')):
    if 'k_call' in r['Name'] or 'k_scan' in r['Name'] or 'k_sum' in r['Name']: print(r['Name'][:40], float(r['AverageNs'])/1000)
PY
grep '^{"metric' gpurun_out/exp.log | cut -c1-100
