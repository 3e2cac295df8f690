#!/bin/bash
# SQ counters of the TIMED k_stream_reads dispatches (the instantiations without the tally, per dispatch -- tools/sq.sh mixes them) of one
# bench configuration, three rocprofv3 --pmc passes: tools/sq_dispatch.sh <tag> <name> <bench args...>
tag=$1; name=$2; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p -o k -- python3 $root/bench.py --steps 20 --warmup 5 --reps 1 --no-cpu-baseline --no-e2e --no-extra "$@" > /dev/null 2> $out/err_${name}_$i.txt
  cp $out/p/k_counter_collection.csv $out/cc_${name}_$i.csv 2>/dev/null
  rm -rf $out/p
done
python3 - <<PY > $out/sq_dispatch_$name.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
order = {}
for f in sorted(glob.glob("$out/cc_${name}_*.csv")):
    seen = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_stream_reads<" not in k: continue
        targs = k[k.index("<") + 1:k.rindex(">")].split(", ")
        if targs[1] != "false": continue            # (the tally pass's instantiation: untimed)
        did = int(r["Dispatch_Id"])
        if did not in seen[k]: seen[k].append(did)
        # (dispatch ids differ from pass to pass: the n-th dispatch of the instantiation is the same launch in every pass)
        acc[(k, seen[k].index(did))][r["Counter_Name"]] += float(r["Counter_Value"])
print("SQ counters of the timed k_stream_reads dispatches, `python3 bench.py --steps 20 --warmup 5 --reps 1 --no-cpu-baseline --no-e2e --no-extra $*`")
print("(three rocprofv3 --kernel-trace --pmc passes; the n-th dispatch of an instantiation is the same launch in each; counters in quad-cycles where they count cycles)")
for (k, n), d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:4]:
    print()
    print(k[k.index("k_stream_reads"):], "dispatch", n)
    for c, v in sorted(d.items()): print("    %-24s %.4e" % (c, v))
    if d.get("SQ_WAVES") and d.get("SQ_WAVE_CYCLES"):
        dur = d["SQ_WAVE_CYCLES"] / d["SQ_WAVES"]
        simds = 1024.0
        print("    -> wave-cycles per wavefront %.0f quad-cycles (persistent wavefronts: the launch's duration); vector instructions per SIMD %.0f (one quad-cycle each) = %.0f %% of it; %.1f wavefronts a SIMD; waiting %.0f %% of wave-cycles"
              % (dur, d["SQ_INSTS_VALU"] / simds, 100.0 * d["SQ_INSTS_VALU"] / simds / dur, d["SQ_WAVES"] / simds, 100.0 * d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]))
PY
cat $out/sq_dispatch_$name.txt
