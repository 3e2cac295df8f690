"""Random differential run of interval sharding: two handles own [0, cut) and [cut, len) of one contig (as two ranks would), the
reads go to the owner of their start, the left handle's halo slab is exported, cleared and added into the right one's planes;
the union of the two handles' rows (keys that both print -- beyond the halo -- summed) must be the oracle's over all reads.
Random cuts and halos (also halos shorter than a read's span), mixed batches, --haplotypes on or off.
usage: python tools/fuzz_shards.py <first seed> <count>"""
import time
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import minimod_amd
from oracle import pybam, oracle as O
from tests import test_hip_stream_gpu as T
from tests.hiprun import to_oracle_rows

first, count = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    ref = T.make_ref(rng, 60000)
    recs = [T._mixed_read(rng, ref) for _ in range(int(rng.integers(20, 100)))]
    c = ("m", "m,h", "m[*],a[*]")[int(rng.integers(0, 3))]
    hap = bool(rng.random() < 0.3)
    if hap:
        for r in recs:
            if rng.random() < 0.6: r.aux += b"HPC" + bytes([int(rng.integers(0, 3))])
    cut = int(rng.integers(8, 90)) * 64
    halo = int(rng.choice([64, 256, 1024, 4096, 16384]))
    mode = int(rng.choice([1, 2, 3]))
    mods = O.parse_mod_codes(c); th = O.parse_mod_threshes(None, len(mods))
    o = O.Oracle(mods, th, ["chrT"], haplotypes=hap); o.add_contig("chrT", ref.encode()); o.process(pybam.flatten(recs))
    wr = o.rows(); wc = o.code_names(); o.close()
    want = {}
    for r in wr: want[(int(r["pos"]), int(r["strand"]), wc[r["code"]], int(r["ins_off"]), int(r["hp"]))] = (int(r["n_called"]), int(r["n_mod"]))
    parts = [[r for r in recs if r.pos < cut], [r for r in recs if r.pos >= cut]]
    ivs = [(0, 0, cut, min(halo, len(ref) - cut)), (0, cut, len(ref), 0)]
    engs = []
    try:
        for k in range(2):
            e = minimod_amd.FreqEngine([(cc, x, t) for (cc, x), t in zip(mods, th)], [("chrT", len(ref), ref.encode())], intervals=[ivs[k]], haplotypes=hap, stream_mode=mode)
            if parts[k]: e.process(pybam.flatten(parts[k]))
            engs.append(e)
        h = ivs[0][3]
        buf = torch.zeros(engs[0].slab_words(h), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()   # (the fill runs on torch's stream, the slab kernels on the handle's own non-blocking one: not ordered otherwise)
        engs[0].slab_export(0, cut, h, buf.data_ptr()); engs[0].slab_clear(0, cut, h); engs[1].slab_add(0, cut, h, buf.data_ptr())
        got = {}
        for e in engs:
            rows = to_oracle_rows(e.finalize()); codes = e.code_names()
            for r in rows:
                key = (int(r["pos"]), int(r["strand"]), codes[r["code"]], int(r["ins_off"]), int(r["hp"]))
                a = got.get(key, (0, 0)); got[key] = (a[0] + int(r["n_called"]), a[1] + int(r["n_mod"]))
            e.close()
        if got != want:
            bad += 1
            d = sorted(set(got.items()) ^ set(want.items()))[:3]
            print("MISMATCH seed", seed, c, "hap", hap, "cut", cut, "halo", halo, "mode", mode, len(got), len(want), d, flush=True)
    except Exception as ex:
        bad += 1; print("ERROR seed", seed, repr(ex)[:200], flush=True)
        for e in engs:
            try: e.close()
            except Exception: pass
print("seeds %d..%d done in %.0f s, %d problems" % (first, first + count - 1, time.time() - t0, bad))
