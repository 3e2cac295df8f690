"""A numpy model of the DEVICE-side replay of the reference's core table and sort (csrc/tie_kernels.hip.h): the same data-parallel
steps, one numpy statement per kernel, checked against the serial restatement in csrc/host/tieorder.c (mmh_tie_order_plain).
It exists to answer two questions before any kernel is written: are the parallel formulations EXACT, and how many rounds do their
fixpoint iterations take.  Nothing in the product imports it.

  khash (reference src/khash.h:242-330, kh_put / kh_resize) in parallel:
    * keys enter a table in first-insertion order, each at the first free slot of its probe path (i += ++step).  In parallel: every
      key of an epoch (the keys between two growths) proposes itself at its current slot with an atomic MIN of its rank; a key that
      finds a smaller rank there moves on; a key whose slot was taken over by a smaller rank later moves on from there.  The fixpoint
      is the serial result: a key sits at the first slot of its path that no EARLIER key holds.
    * a growth rehashes in place with kick-outs: bucket j's key goes to its slot of the new table, and if that slot (as a bucket of the
      old table) holds a key that has not moved yet, that key goes next.  The new table is therefore "insert into an empty table in
      chain order", and the chain order depends on where keys land.  Land := home slot gives a first order; the placement under that
      order gives new landings; iterate until nothing moves (the pair (order, landing) is unique, by induction along the order).
  ks_introsort (reference src/ksort.h:180-230) in parallel:
    * its partition loop swaps the k-th element from the left that is not smaller than the pivot with the k-th from the right that is
      not bigger, while they have not crossed: ranks by prefix sums, all swaps at once; segments of one level side by side;
    * the insertion sort that ends it is a stable sort of what the partitions left.

usage: python tools/tie_model.py [n] [seeds]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)


def host_lib():
    from minimod_amd import build as B
    L = ctypes.CDLL(os.path.join(B.LIBDIR, "libminimod_host.so"))
    L.mmh_tie_order_plain.restype = ctypes.c_int
    L.mmh_tie_order_plain.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return L


def host_order(hash32, sortkey, put_after_last=0):
    L = host_lib()
    n = len(hash32)
    slot = np.zeros(n, np.uint32)
    fin = np.zeros(n, np.uint32)
    h = np.ascontiguousarray(hash32, np.uint32)
    k = np.ascontiguousarray(sortkey, np.int64)
    assert L.mmh_tie_order_plain(h.ctypes.data, k.ctypes.data, n, put_after_last, slot.ctypes.data, fin.ctypes.data) == 0
    return slot, fin


def upper(c):
    return int(c * 0.77 + 0.5)


def place(table, words, home, mask, stats):
    """every key at the first slot of its path that no smaller word holds; `table` holds the words already there (smaller than all of `words`)"""
    n = len(words)
    cur = home.copy()
    step = np.zeros(n, np.int64)
    rounds = 0
    while True:
        rounds += 1
        # a kernel: every key that does not hold its slot proposes / moves on until it holds one or meets a bigger word
        active = np.nonzero(table[cur] != words)[0]
        if len(active) == 0:
            break
        moving = active
        while len(moving):
            # atomicMin of the word at the key's slot; the loser (old value smaller) moves on
            np.minimum.at(table, cur[moving], words[moving])
            lost = table[cur[moving]] < words[moving]   # (a serial model: "smaller than me there now" is the same test)
            moving = moving[lost]
            step[moving] += 1
            cur[moving] = (cur[moving] + step[moving]) & mask
    stats["place_rounds"] = max(stats.get("place_rounds", 0), rounds)
    return cur


def grow(old, hash32, stats):
    """old: table of ranks (uint64, EMPTY) with C buckets -> the table of 2C buckets the in-place rehash leaves"""
    C = len(old)
    C2 = 2 * C
    mask = C2 - 1
    occ = np.nonzero(old != EMPTY)[0]                 # occupied old buckets, ascending
    ranks = old[occ].astype(np.int64)
    home = (hash32[ranks].astype(np.int64)) & mask
    land = home.copy()
    slot_of = np.full(C, -1, np.int64)                # old bucket -> index into occ
    slot_of[occ] = np.arange(len(occ))
    it = 0
    while True:
        it += 1
        # successor: the key in the old bucket this key lands on (not itself)
        tgt = np.where(land < C, land, -1)
        succ = np.where(tgt >= 0, slot_of[np.maximum(tgt, 0)], -1)
        succ = np.where(succ == np.arange(len(occ)), -1, succ)
        pred = np.full(len(occ), -1, np.int64)
        has = succ >= 0
        pred[succ[has]] = np.nonzero(has)[0]
        # walk back along the predecessors: the smallest bucket met and how far back it lies (paths are short: a landing continues a
        # chain with probability ~0.39)
        best = occ.copy()
        dist = np.zeros(len(occ), np.int64)
        p = pred.copy()
        d = np.ones(len(occ), np.int64)
        me = np.arange(len(occ))
        steps = 0
        while True:
            live = (p >= 0) & (p != me)
            if not live.any():
                break
            steps += 1
            better = live & (occ[np.maximum(p, 0)] < best)
            best = np.where(better, occ[np.maximum(p, 0)], best)
            dist = np.where(better, d, dist)
            p = np.where(live, pred[np.maximum(p, 0)], -1)
            d += 1
            if steps > 4096:
                raise RuntimeError("chain too long")
        stats["chain_steps"] = max(stats.get("chain_steps", 0), steps)
        words = (best.astype(np.uint64) << np.uint64(32)) | (dist.astype(np.uint64) << np.uint64(0))
        # (unique per key: a chain start's bucket and the distance from it name one key; the key's own bucket is not needed in the word here
        # because the model keeps `cur` per key)
        new = np.full(C2, EMPTY, np.uint64)
        new_land = place(new, words, home, mask, stats)
        if (new_land == land).all():
            break
        land = new_land
        if it > 200:
            raise RuntimeError("no consistent order after 200 iterations")
    stats["grow_iters"] = max(stats.get("grow_iters", 0), it)
    stats.setdefault("grow_iters_all", []).append(it)
    out = np.full(C2, EMPTY, np.uint64)
    out[land] = ranks.astype(np.uint64)
    return out


def core_table(hash32, put_after_last, stats):
    n = len(hash32)
    C = 4
    table = np.full(C, EMPTY, np.uint64)
    done = 0
    while True:
        U = upper(C)
        hi = min(n, U)
        if hi > done:
            ranks = np.arange(done, hi, dtype=np.int64)
            place(table, ranks.astype(np.uint64), hash32[ranks].astype(np.int64) & (C - 1), C - 1, stats)
            done = hi
        if done == n:
            if put_after_last and n >= U:
                table = grow(table, hash32, stats)
            break
        table = grow(table, hash32, stats)
        C *= 2
    return table[table != EMPTY].astype(np.int64)


def lt(a, b):
    return a < b


def partition_level(keys, ids, segs, stats):
    """one level: every segment (s, t, d) partitioned as ksort.h does; returns the child segments"""
    out = []
    for (s, t, d) in segs:   # (side by side on the device: the segments are disjoint)
        d -= 1
        if d == 0:
            raise RuntimeError("depth budget spent: comb sort (not modelled)")
        k = s + ((t - s) >> 1) + 1
        if keys[k] < keys[s]:
            if keys[k] < keys[t]:
                k = t
        else:
            k = s if keys[t] < keys[s] else t
        rp = keys[k]
        if k != t:
            keys[k], keys[t] = keys[t], keys[k]
            ids[k], ids[t] = ids[t], ids[k]
        seg = keys[s:t + 1]
        idx = np.arange(s, t + 1)
        Lf = (idx > s) & ~(seg < rp)
        Rf = (idx > s) & (idx < t) & ~(rp < seg)
        Lpos = idx[Lf]                      # k-th from the left
        Rpos = idx[Rf][::-1]                # k-th from the right
        m = min(len(Lpos), len(Rpos))
        sw = Lpos[:m] < Rpos[:m]
        K = int(sw.sum())
        assert sw[:K].all()
        a, b = Lpos[:K], Rpos[:K]
        keys[a], keys[b] = keys[b].copy(), keys[a].copy()
        ids[a], ids[b] = ids[b].copy(), ids[a].copy()
        i = int(min(Lpos[K], Rpos[K - 1] if K > 0 else t))
        keys[i], keys[t] = keys[t], keys[i]
        ids[i], ids[t] = ids[t], ids[i]
        if i - s > 16:
            out.append((s, i - 1, d))
        if t - i > 16:
            out.append((i + 1, t, d))
    return out


def intro_sort(keys, ids, stats):
    n = len(keys)
    if n < 1:
        return
    if n == 2:
        if keys[1] < keys[0]:
            keys[[0, 1]] = keys[[1, 0]]
            ids[[0, 1]] = ids[[1, 0]]
        return
    d = 2
    while (1 << d) < n:
        d += 1
    segs = [(0, n - 1, d << 1)] if n - 1 > 0 else []
    # ksort.h partitions ANY segment with s < t that it is handed; only CHILDREN are cut off at 16
    levels = 0
    while segs:
        levels += 1
        segs = partition_level(keys, ids, segs, stats)
    stats["sort_levels"] = levels
    order = np.argsort(keys, kind="stable")
    keys[:] = keys[order]
    ids[:] = ids[order]


def model_order(hash32, sortkey, put_after_last=0):
    stats = {}
    slot = core_table(np.asarray(hash32, np.uint32), put_after_last, stats)
    keys = np.asarray(sortkey, np.int64)[slot].copy()
    ids = slot.copy()
    intro_sort(keys, ids, stats)
    return slot.astype(np.uint32), ids.astype(np.uint32), stats


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    for seed in range(seeds):
        rng = np.random.default_rng(seed)
        h = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        # sites with two to four rows each, as a two-code run has them
        sk = np.sort(rng.integers(0, max(2, n // 3), n)).astype(np.int64)
        sk = sk[rng.permutation(n)]
        for pal in (0, 1):
            hs, hf = host_order(h, sk, pal)
            ms, mf, st = model_order(h, sk, pal)
            ok1, ok2 = bool((hs == ms).all()), bool((hf == mf).all())
            print("n %d seed %d put_after_last %d: core table %s, printed order %s, %s" % (n, seed, pal, "==" if ok1 else "DIFFERS", "==" if ok2 else "DIFFERS",
                  {k: v for k, v in st.items() if k != "grow_iters_all"}), st.get("grow_iters_all"))
            assert ok1 and ok2


if __name__ == "__main__":
    main()
