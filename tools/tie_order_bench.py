"""The device's tie ordering alone (mm_tie_order_plain: the core hash table's slot order, then ks_introsort's order) on n keys shaped like a C3 run's
-- hashes of distinct keys, sort keys = positions with two rows a position -- with MM_TIMELINE=1 the phases' milliseconds.
usage: MM_TIMELINE=1 python tools/tie_order_bench.py [n] [reps]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from minimod_amd import tie as TT

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(1)
L = TT._lib()
hash_ = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
pos = np.sort(rng.integers(0, 50_000_000, size=n // 2 + 1))[:(n + 1) // 2]
sortkey = np.repeat(pos, 2)[:n].astype(np.int64)
order = rng.permutation(n)                      # first-insertion order: reads arrive in position order only roughly
sortkey = sortkey[order]
final = np.zeros(n, dtype=np.uint32)
stats = (ctypes.c_uint64 * 8)()
for r in range(reps):
    t0 = time.perf_counter()
    rc = L.mm_tie_order_plain(0, hash_.ctypes.data, sortkey.ctypes.data, n, 0, None, final.ctypes.data)
    dt = time.perf_counter() - t0
    L.mm_tie_last_stats(stats)
    assert rc == 0, rc
    print("n %d: %.1f ms wall; %d launches, %d growths, %d passes, %d rounds, %d levels, %d small segments" % (n, 1e3 * dt, stats[0], stats[1], stats[2], stats[3], stats[4], stats[5]), flush=True)
# the order is a stable sort by key of the slot order's partition leftovers: at least sorted by key
assert (np.diff(sortkey[final].astype(np.int64)) >= 0).all()
