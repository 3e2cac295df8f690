cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmcq -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 $BARGS > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/pmcq/*/*counter_collection.csv'))[-1]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'].split('(')[0].split('::')[-1].split('<')[0]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if not k.startswith('k_'): continue
    m={c:sum(v[-10:])/len(v[-10:]) for c,v in d.items()}
    print(k, 'waves %d valu/wave %.0f salu/wave %.0f lds/wave %.0f wavecyc/wave(x4) %.0f wait_any %.0f%% wait_inst %.0f%% active %.0f%%' % (m['SQ_WAVES'], m['SQ_INSTS_VALU']/m['SQ_WAVES'], m['SQ_INSTS_SALU']/m['SQ_WAVES'], m['SQ_INSTS_LDS']/m['SQ_WAVES'], 4*m['SQ_WAVE_CYCLES']/m['SQ_WAVES'], 100*m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES'], 100*m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES'], 100*m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']))
PY
rm -rf gpurun_out/pmcq
