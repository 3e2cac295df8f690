#!/bin/bash
timeout 200 python -m pytest tests/test_hip_bgzf_gpu.py -x -q --timeout 100 2>&1 | tail -3
timeout 120 python - <<'PY'
import sys, time, os, zlib
sys.path.insert(0,'.')
import numpy as np
from minimod_amd import bgzf, synth
inf = bgzf.Inflater(slots=1, max_blocks=8192, max_cbytes=200 << 20, max_obytes=600 << 20)
ref = synth.reference(3, 16 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=4*4096, with_order=False) for i in range(4)]
synth.write_bam_parallel("/tmp/s.bam", [("chrS", len(ref))], bs, threads=8)
data = open("/tmp/s.bam","rb").read()
blocks = bgzf.split_bgzf(data)
n = min(len(blocks), 8192)
nb, c, o = inf.fill(0, blocks[:n])
for rep in range(2):
    inf.submit(0, nb, c, o); st = inf.wait(0, nb)
    t = inf.times(0)
    print("bad", int((st != 0).sum()), t, "inflate GB/s decoded %.1f" % (o / t["inflate_ms"] / 1e6), flush=True)
# 1024-block launches, as the CLI makes them
nb, c, o = inf.fill(0, blocks[:1024])
inf.submit(0, nb, c, o); st = inf.wait(0, nb); t = inf.times(0)
print("1024 blocks:", t, "GB/s %.1f" % (o / t["inflate_ms"] / 1e6))
inf.close()
PY
