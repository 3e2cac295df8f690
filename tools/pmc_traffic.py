#!/usr/bin/env python3
"""HBM traffic of the freq hot path from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, see
tools/profile_round.sh): the LAST dispatch of every hot-path kernel is the timed launch of the profiled command (its
--steps batches gathered into one launch); counters are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section)
prescribes for gfx950.  usage: pmc_traffic.py FETCH.csv WRITE.csv <batches in the timed launch> <algorithmic bytes per batch> > profiles/traffic_c2.json"""
import csv, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minimod_amd.build import source_hash
KERNELS = ("k_plan_items", "k_stream_reads", "k_scan_reads", "k_sum_tiles", "k_call_tiles", "k_freq_reads", "k_view_offsets", "k_view_offsets_apply", "k_view_scatter", "k_view_sort", "k_view_compact")
def last(path):
    out = {}
    for r in csv.DictReader(open(path)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                out[k] = float(r["Counter_Value"])
    return out
f, w = last(sys.argv[1]), last(sys.argv[2])
nb, alg = int(sys.argv[3]), float(sys.argv[4])
label = sys.argv[5] if len(sys.argv) > 5 else "workload C2"
KERNELS = tuple(k for k in KERNELS if k in f and k in w)
per_launch = sum(2 * f[k] + w[k] for k in KERNELS) * 1024
print(json.dumps({
    "source_hash": source_hash(),
    "what": "HBM traffic of the hot path's kernels in the timed launch (%s), %s, one launch of %d gathered -K 4096 batches" % (" + ".join(KERNELS), label, nb),
    "how": "two separate rocprofv3 passes, `--kernel-trace --pmc FETCH_SIZE` and `--kernel-trace --pmc WRITE_SIZE` (never combined, no sys/hip trace), on "
           "`python3 bench.py --steps %d --warmup 0 --no-cpu-baseline --no-e2e --no-extra`; counters are in KB; per MI355X_MICROARCH.md (HBM section) "
           "FETCH_SIZE on gfx950 reports half of the bytes fetched, so it is doubled: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (checked on this access mix with tools/fetch_calib.hip: FETCH_SIZE = half of 128 B per line touched for 16-, 4- and 1-byte streaming loads and for 16- and 2-byte gathers alike; WRITE_SIZE exact for stores, 32 B per scattered 64-bit atomic)" % nb,
    "raw_kb_per_launch": {k: {"FETCH_SIZE": f[k], "WRITE_SIZE": w[k]} for k in KERNELS},
    "batches_per_launch": nb,
    "hbm_bytes_per_launch": per_launch,
    "hbm_bytes_per_batch": per_launch / nb,
    "algorithmic_bytes_per_batch": alg,
    "ratio": per_launch / nb / alg,
}, indent=1))
