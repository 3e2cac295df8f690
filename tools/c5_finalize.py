#!/usr/bin/env python3
"""C5 (--haplotypes --insertions, 200x on 5 Mb): time of the batches and of mm_freq_finalize, for the device library given
by MM_HIP_LIB (default: the in-tree build).  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import minimod_amd
from minimod_amd import synth
region = 5 << 20
ref = synth.reference(31, region + (1 << 20))
n = 66000
bs = [synth.batch(ref, i * 4096, min(4096, n - i * 4096), seed=41, n_reads_total=n, region_begin=0, region_len=region, haplotypes=True, long_insertions=True)
      for i in range((n + 4095) // 4096)]
bases = sum(b["n_bases"] for b in bs)
eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)], insertions=True, haplotypes=True, side_capacity=96 << 20)
for rep in range(2):
    eng.reset()
    t0 = time.time(); tk = []
    for b in bs:
        tk.append(eng.submit(b))
        if len(tk) >= 3: eng.wait(tk.pop(0))
    for t in tk: eng.wait(t)
    t1 = time.time()
    rows = eng.finalize()
    t2 = time.time()
eng.close()
print(json.dumps({"lib": os.environ.get("MM_HIP_LIB", "in-tree"), "reads": n, "bases": int(bases), "rows": int(len(rows)), "batches_s": t1 - t0, "finalize_s": t2 - t1,
                  "rows_with_ins_offset": int((rows["ins_offset"] > 0).sum())}))
