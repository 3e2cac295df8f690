"""Random differential run with LONG reads and LONG CIGARs: window overflows of k_stream_reads (more than 576 ops / 320 blocks
per round), reads beyond split_bases (parts in the tile pipeline), dense and sparse lists.  usage: tools/fuzz_long.py <first> <count>"""
import time
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pybam
from tests import test_hip_stream_gpu as T

def lread(rng, ref):
    flag = 16 if rng.random() < 0.5 else 0
    n_ops = int(rng.choice([10, 100, 700, 2500]))
    mean = float(rng.choice([6, 15, 40]))
    pos = int(rng.integers(0, 5000))
    rp, ops, seq = pos, [], []
    if rng.random() < 0.3:
        l = int(rng.integers(1, 200)); ops.append("%dS" % l); seq.append(T.make_ref(rng, l))
    for i in range(n_ops):
        l = int(rng.geometric(1.0 / mean))
        if rng.random() < 0.01: l = int(rng.integers(1000, 20000))
        kind = "M" if i == 0 or i == n_ops - 1 else str(rng.choice(list("MMMMM=XIDN")))
        if rp + l >= len(ref) - 1000: break
        if kind in "M=X":
            seq.append(ref[rp:rp + l]); rp += l
        elif kind == "I":
            seq.append(T.make_ref(rng, l))
        else:
            rp += l
        ops.append("%d%s" % (l, kind))
    if ops[-1][-1] in "IDN":
        ops.append("5M"); seq.append(ref[rp:rp + 5]); rp += 5
    seq = "".join(seq)
    orig = T.revcomp(seq) if flag else seq
    n_c = orig.count("C")
    mm, ml = "", []
    for g in range(int(rng.integers(1, 4))):
        codes = "".join(rng.permutation(list("mhx"))[:int(rng.integers(1, 3))])
        fl = str(rng.choice(["?", "?", ".", ""]))
        dens = float(rng.choice([0.001, 0.02, 0.3, 1.0]))
        picks = np.nonzero(rng.random(n_c) < dens)[0].tolist()
        toks, prev = [], -1
        for k in picks:
            toks.append(str(k - prev - 1)); prev = k
        mm += "C+%s%s" % (codes, fl) + "".join("," + t for t in toks) + ";"
        ml += [int(x) for x in rng.integers(0, 256, size=len(toks) * len(codes))]
    return pybam.make_record(0, pos, flag, seq, "".join(ops), mm, ml)

first, count = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    ref = T.make_ref(rng, 400000)
    recs = [lread(rng, ref) for _ in range(int(rng.integers(3, 30)))]
    c = ("m", "m,h", "m[*],h[*]")[int(rng.integers(0, 3))]
    want = T.oracle_rows(recs, ref, c)
    for kw in (dict(stream_mode=3), dict(stream_mode=2), dict(stream_mode=1), dict(stream_mode=3, split_bases=4096), dict(stream_mode=1, split_bases=2048)):
        got, st = T.hip_rows(recs, ref, c, **kw)
        if got != want:
            bad += 1; print("MISMATCH seed", seed, c, kw, len(got), len(want), sorted(set(got) ^ set(want))[:3], st, flush=True)
print("seeds %d..%d done in %.0f s, %d problems; longest read %d" % (first, first + count - 1, time.time() - t0, bad, max(r.l_qseq for r in recs)))
