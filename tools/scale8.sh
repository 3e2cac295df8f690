#!/bin/bash
# scale8.sh -- the day an 8-GPU node is there: the scaling curve of BASELINE configs[3] (C4: 30x whole genome, contig-sharded) in one command.
#   tools/scale8.sh [N=8] [backend=nccl] [gbases=93]
# 1. bench.py --config C4 --gpus N : every rank a 388 Mb share of the 24-contig genome and 775 000 reads resident in HBM; the line carries
#    per-rank Mbases/s and roofline fractions, final_reduce {ms, ranks_seen, slab_bytes, backend} (halo slabs over RCCL), value = all bases /
#    the slowest rank's time.  Also N = 1, 2, 4 when N = 8, so that the curve is in one directory.
# 2. bench.py --e2e-gbases G --e2e-devices 0,...,N-1 : ONE BAM of G Gbases through `minimod freq --devices` (one worker process per GPU reading
#    its share through the .bai, halo slabs from GPU to GPU through HIP IPC handles -- "slabs_through_hip_ipc" in the line says which transport
#    ran --, sections of text put in order by the parent), bytes compared with the single run's.
# A dry run on one GPU: tools/scale8.sh 2 gloo 1   (both ranks share the GPU: it shows the path, not scaling).
set -e
cd "$(dirname "$0")/.."
N=${1:-8}; BACKEND=${2:-nccl}; G=${3:-93}
OUT=gpurun_out/scale8; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NS="$N"; [ "$N" = 8 ] && NS="1 2 4 8"
for n in $NS; do
  if [ "$n" = 1 ]; then python bench.py --config C4 --gpus 1 --steps 20 --warmup 5 --no-e2e --no-config-fracs > "$OUT/C4_n1.json" 2> "$OUT/C4_n1.err"
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --config C4 --gpus "$n" --steps 20 --warmup 5 --backend "$BACKEND" --no-e2e --no-config-fracs > "$OUT/C4_n$n.json" 2> "$OUT/C4_n$n.err"
  fi
  python - "$OUT/C4_n$n.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fr = d.get("final_reduce") or {}
print("C4 n=%d: %.0f Mbases/s, frac %.3f, final_reduce %s ms over %s ranks (%s), per rank %s" % (d["n_gpus"], d["value"], d["roofline"]["frac"], fr.get("ms"), fr.get("ranks_seen"), fr.get("backend"),
      [round(p.get("value", 0)) for p in (d.get("per_rank") or [])]))
PY
done
DEVS=$(python -c "print(','.join(str(i % max(1, __import__('torch').cuda.device_count())) for i in range($N)))")
python bench.py --e2e-gbases "$G" --e2e-devices "$DEVS" > "$OUT/e2e_devices_n$N.json" 2> "$OUT/e2e_devices_n$N.err"
python - "$OUT/e2e_devices_n$N.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["devices_run"]
print("--devices %s: %.2f s (single run %.2f s), %d slabs through HIP IPC, byte-identical to the single run: %s, workers %s" % (r["devices"], r["wall_s"], d["gpu_cli"]["wall_s"], r["slabs_through_hip_ipc"],
      r["byte_identical_to_single_run"], [(w["device"], w["mbases"]) for w in r["workers"]]))
PY
