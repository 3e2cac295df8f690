"""Random differential run, wider shapes: '.' / '?' / no flag per group, groups on other bases, unrequested codes between
requested ones, zero-padded tokens, both strands, reads with several groups on the same base.  HIP (stream '.', stream lean,
tiles, fused) vs the oracle.  usage: python tools/fuzz_mixed.py <first seed> <count>"""
import time
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pybam
from tests import test_hip_stream_gpu as T

def rread(rng, ref):
    # a third of the reads carry twin groups (two one-code `?` lists over the same tokens) in their variants
    if rng.random() < 0.33:
        v = str(rng.choice(["same", "same", "padded", "differs", "shorter", "empty_token", "third"]))
        return T._twin_read(rng, ref, int(rng.integers(0, 3000)), int(rng.integers(50, 6000)), 16 if rng.random() < 0.5 else 0, v)
    return T._mixed_read(rng, ref)

first, count = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    ref = T.make_ref(rng, 120000)
    recs = [rread(rng, ref) for _ in range(int(rng.integers(10, 120)))]
    for c in [("m", "m,h", "m[*],h[*]", "h[CG]", "m[C]", "x[*]", "m,h,x", "a[A]", "m[CG],a[*]")[int(rng.integers(0, 9))] for _ in range(2)]:
        try:
            want = T.oracle_rows(recs, ref, c)
        except Exception as e:
            want = ("oracle error", str(e)[:80])
        for kw in (dict(stream_mode=3), dict(stream_mode=2), dict(stream_mode=1), dict(force_fused=True)):
            try:
                got, st = T.hip_rows(recs, ref, c, **kw)
            except Exception as e:
                got = ("hip error", str(e)[:80])
            if isinstance(want, tuple) != isinstance(got, tuple) or (not isinstance(want, tuple) and got != want):
                bad += 1
                print("MISMATCH seed", seed, c, kw, str(got)[:100], str(want)[:100], flush=True)
print("seeds %d..%d done in %.0f s, %d problems" % (first, first + count - 1, time.time() - t0, bad))
