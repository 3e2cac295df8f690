"""Random differential run over several contigs: names that sort differently by strcmp than by tid, lengths from 3 kb to 90 kb
(contig starts in the reference array at odd multiples of 64), a contig that is in the BAM header but not in the FASTA and
carries no reads, mixed reads on all the others, several batches; rows compared IN ORDER with tid.
usage: python tools/fuzz_contigs.py <first seed> <count>"""
import time
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pybam, oracle as O
from tests import test_hip_stream_gpu as T
from tests.hiprun import make_engine, to_oracle_rows

NAMES = ["chr1", "chr10", "chr2", "chrX", "chrM", "chr1_alt", "2", "10", "GL0001.1", "chrUn"]
first, count = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 7))
    names = [str(x) for x in rng.permutation(NAMES)[:n]]
    lens = [int(rng.integers(3000, 90000)) for _ in range(n)]
    refs = [T.make_ref(rng, l) for l in lens]
    absent = int(rng.integers(0, n)) if rng.random() < 0.5 else -1
    recs = []
    for _ in range(int(rng.integers(20, 120))):
        t = int(rng.integers(0, n))
        if t == absent: continue
        for _try in range(20):
            r = T._mixed_read(rng, refs[t])
            span = sum(int(x) >> 4 for x in r.cigar if (int(x) & 15) in (0, 2, 3, 7, 8))
            if r.pos + span <= lens[t]: break
        else:
            continue
        r.tid = t; recs.append(r)
    if not recs: continue
    c = ("m", "m,h", "m[*],a[*]", "h[CG]")[int(rng.integers(0, 4))]
    ins, hap = bool(rng.random() < 0.25), bool(rng.random() < 0.25)
    mods = O.parse_mod_codes(c); th = O.parse_mod_threshes(None, len(mods))
    nb = int(rng.integers(1, 4)); cut = sorted(rng.integers(0, len(recs) + 1, size=nb - 1).tolist())
    batches = [b for b in (recs[a:b] for a, b in zip([0] + cut, cut + [len(recs)])) if b]
    o = O.Oracle(mods, th, names, insertions=ins, haplotypes=hap)
    for t in range(n):
        if t != absent: o.add_contig(names[t], refs[t].encode())
    for b in batches: o.process(pybam.flatten(b))
    wr = o.rows(); wc = o.code_names(); o.close()
    want = [(int(r["tid"]), int(r["pos"]), int(r["strand"]), wc[r["code"]], int(r["ins_off"]), int(r["hp"]), int(r["n_called"]), int(r["n_mod"])) for r in wr]
    for kw in (dict(stream_mode=3), dict(stream_mode=1), dict(force_fused=True), dict(stream_mode=2, coalesce=4)):
        try:
            eng = make_engine(mods, th, names, lens, {names[t]: refs[t].encode() for t in range(n) if t != absent}, insertions=ins, haplotypes=hap, **kw)
            for b in batches: eng.process(pybam.flatten(b))
            rows = to_oracle_rows(eng.finalize()); codes = eng.code_names(); eng.close()
            got = [(int(r["tid"]), int(r["pos"]), int(r["strand"]), codes[r["code"]], int(r["ins_off"]), int(r["hp"]), int(r["n_called"]), int(r["n_mod"])) for r in rows]
        except Exception as ex:
            got = ("error", repr(ex)[:120])
        if got != want:
            bad += 1
            same_set = not isinstance(got, tuple) and sorted(got) == sorted(want)
            print("MISMATCH seed", seed, names, c, "ins", ins, "hap", hap, kw, "same rows, other order" if same_set else str(got)[:100], flush=True)
print("seeds %d..%d done in %.0f s, %d problems" % (first, first + count - 1, time.time() - t0, bad))
