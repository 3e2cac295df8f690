cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_hip_view_gpu.py -x -q -m gpu 2>&1 | tail -1
for m in freq view; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp_$m -o e -- python3 bench.py --mode $m --no-cpu-baseline --no-extra --steps 30 > gpurun_out/exp_$m.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/exp_$m/e_kernel_stats.csv')):
    if "k_call" in r["Name"] or "k_scan" in r["Name"]: print('$m', r['Name'][:50], float(r['AverageNs'])/1000)
PY
grep '^{"metric' gpurun_out/exp_$m.log | cut -c1-90
done
