"""Random differential run over the options: --insertions / --haplotypes (general call kernel, side table), view rows,
several batches per handle, long reads cut into parts (small split_bases).  usage: python tools/fuzz_options.py <first seed> <count>"""
import time
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pybam, oracle as O
from tests import test_hip_stream_gpu as T
from tests.hiprun import make_engine, to_oracle_rows

def key_rows(rows, codes):
    return [(int(r["pos"]), "+-"[r["strand"]], int(r["n_called"]), int(r["n_mod"]), int(r["ins_off"]), int(r["hp"]), codes[r["code"]]) for r in rows]

def run_freq(batches, ref, c, ins, hap, **kw):
    mods = O.parse_mod_codes(c)
    eng = make_engine(mods, O.parse_mod_threshes(None, len(mods)), ["chrT"], [len(ref)], {"chrT": ref.encode()}, insertions=ins, haplotypes=hap, **kw)
    for b in batches:
        eng.process(pybam.flatten(b))
    rows = to_oracle_rows(eng.finalize()); codes = eng.code_names(); eng.close()
    return key_rows(rows, codes)

def orc_freq(batches, ref, c, ins, hap):
    mods = O.parse_mod_codes(c)
    o = O.Oracle(mods, O.parse_mod_threshes(None, len(mods)), ["chrT"], insertions=ins, haplotypes=hap)
    o.add_contig("chrT", ref.encode())
    for b in batches:
        o.process(pybam.flatten(b))
    rows = o.rows(); codes = o.code_names(); o.close()
    return key_rows(rows, codes)

def run_view(recs, ref, c, ins, hap, **kw):
    mods = O.parse_mod_codes(c)
    eng = make_engine(mods, O.parse_mod_threshes(None, len(mods)), ["chrT"], [len(ref)], {"chrT": ref.encode()}, view=True, insertions=ins, haplotypes=hap, **kw)
    rows = eng.view(pybam.flatten(recs)); codes = eng.code_names(); eng.close()
    return [(int(r["read"]), int(r["pos"]), int(r["read_pos"]), codes[r["code"]], int(r["prob"]), int(r["ins_offset"])) for r in rows]

def orc_view(recs, ref, c, ins, hap):
    mods = O.parse_mod_codes(c)
    o = O.Oracle(mods, O.parse_mod_threshes(None, len(mods)), ["chrT"], insertions=ins, haplotypes=hap)
    o.set_view(True); o.add_contig("chrT", ref.encode()); o.process(pybam.flatten(recs))
    codes = o.code_names()
    out = [(int(r["read"]), int(r["pos"]), int(r["read_pos"]), codes[r["code"]], int(r["prob"]), int(r["ins_off"])) for r in o.view_rows()]
    o.close(); return out

first, count = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    ref = T.make_ref(rng, 120000)
    recs = [T._mixed_read(rng, ref) for _ in range(int(rng.integers(20, 110)))]
    if rng.random() < 0.5:
        for r in recs:   # haplotype tags on some reads
            if rng.random() < 0.6:
                r.aux += b"HPC" + bytes([int(rng.integers(0, 4))])
    c = ("m", "m,h", "m[*],a[*]", "h[CG]", "m[C],x[*]", "*", "*[C]")[int(rng.integers(0, 7))]   # (the last two: every code the reads carry)
    ins, hap = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
    nb = int(rng.integers(1, 4))
    cut = sorted(rng.integers(0, len(recs) + 1, size=nb - 1).tolist())
    batches = [b for b in (recs[a:b] for a, b in zip([0] + cut, cut + [len(recs)])) if b]
    try:
        want = orc_freq(batches, ref, c, ins, hap)
        for kw in (dict(stream_mode=3), dict(stream_mode=1), dict(stream_mode=1, split_bases=1024), dict(force_fused=True), dict(stream_mode=2, coalesce=4)):
            got = run_freq(batches, ref, c, ins, hap, **kw)
            if got != want:
                bad += 1; print("FREQ MISMATCH seed", seed, c, "ins", ins, "hap", hap, kw, len(got), len(want), sorted(set(got) ^ set(want))[:3], flush=True)
        wantv = orc_view(recs, ref, c, ins, hap)
        for kw in (dict(stream_mode=3), dict(stream_mode=1), dict(force_fused=True)):
            gotv = run_view(recs, ref, c, ins, hap, **kw)
            if gotv != wantv:
                bad += 1; print("VIEW MISMATCH seed", seed, c, "ins", ins, "hap", hap, kw, len(gotv), len(wantv), sorted(set(gotv) ^ set(wantv))[:3], flush=True)
    except Exception as e:
        bad += 1; print("ERROR seed", seed, c, ins, hap, repr(e)[:200], flush=True)
print("seeds %d..%d done in %.0f s, %d problems" % (first, first + count - 1, time.time() - t0, bad))
