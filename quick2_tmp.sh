bash quick_tmp.sh
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -- python3 bench.py --no-cpu-baseline --no-extra --steps 25 --warmup 2 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_q/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
for name in ('k_scan_reads','k_sum_tiles','k_call_tiles','k_freq_reads'):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in rows if name in r['Kernel_Name']][-25:]
    if d: print(name, 'mean', round(sum(d)/len(d),1), 'min', round(min(d)), 'max', round(max(d)))
PY
rm -rf gpurun_out/prof_q
