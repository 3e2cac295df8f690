cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --no-cpu-baseline --no-extra --steps 25 --warmup 2 > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,collections,json
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=sorted(glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv'%c))[-1]
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        k=r['Kernel_Name'].split('(')[0].split('::')[-1].split('<')[0]
        if r['Counter_Name']==c: agg[k].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if k.startswith('k_') and k!='k_build_refwords':
            v=v[-25:]   # the timed steps
            out.setdefault(k,{})[c]=sum(v)/len(v)
print(json.dumps(out))
tot_f=sum(d.get('FETCH_SIZE',0) for d in out.values()); tot_w=sum(d.get('WRITE_SIZE',0) for d in out.values())
print('per batch: FETCH_SIZE %.1f KB  WRITE_SIZE %.1f KB  -> hbm bytes (2*FETCH+WRITE)*1024 = %.1f MB' % (tot_f, tot_w, (2*tot_f+tot_w)*1024/1e6))
PY
