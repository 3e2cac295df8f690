"""Host-side cost of the tie-order replay (csrc/host/tieorder.c) without a GPU: synthetic C3 batches (two codes: every site ties), the
calls from the oracle's view mode 2 as the device would deliver them, timed through mmh_tie_add_batch and mmh_tie_order_rows.
usage: python tools/tie_bench.py [reads] [threads]   -- prints seconds and an md5 of the resulting row order (to compare builds)"""
import ctypes
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle as O
from tests.test_tieorder_cpu import _lib


def main():
    from minimod_amd.engine import ROW_DTYPE, VIEW_ROW_DTYPE, batch_struct
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    L, hostlib = _lib()
    wl = bench.WORKLOADS["C3"]
    plan = bench.shard_plan(0, 1, interval=max(1 << 20, n_reads * 512))
    ref = bench.gen_reference(plan, 7)
    mods = [(c, x) for c, x, _ in wl["mods"]]
    th = [t for _, _, t in wl["mods"]]
    luts = [hostlib.klass_lut(t) for t in th]
    names = ["chrS"]
    cnt = O.Oracle(mods, th, names); cnt.add_contig("chrS", ref)
    tn = (ctypes.c_char_p * 1)(b"chrS"); tl = (ctypes.c_uint32 * 1)(len(ref))
    hdr = hostlib.mm_bam_hdr_t(1, tn, tl)
    tie = L.mmh_tie_create(ctypes.byref(hdr), 0, 0)
    pool = L.mm_pool_create(threads)
    t_add = 0.0
    n_rows = 0
    for bi in range((n_reads + 4095) // 4096):
        b = bench.gen_batch(ref, plan, 0, 7, n_reads, 4096, bi, **wl["gen"])
        cnt.process(b, threads)
        vw = O.Oracle(mods, th, names); vw.set_view(2); vw.add_contig("chrS", ref); vw.process(b, threads)
        v = vw.view_rows()
        rows = np.zeros(len(v), dtype=VIEW_ROW_DTYPE)
        rows["read"] = v["read"].astype(np.uint32) | (((v["prob"] >> 8) & 0xFF).astype(np.uint32) << 21)
        rows["pos"], rows["ins_offset"], rows["code"] = v["pos"], v["ins_off"], v["code"]
        rows["read_pos"] = v["read_pos"].astype(np.uint32) | (v["prob"] & 0x80000000).astype(np.uint32)
        rows["prob"] = (v["prob"] & 0xFF).astype(np.uint8)
        rows = np.ascontiguousarray(rows[np.lexsort((rows["ins_offset"], rows["code"], rows["pos"], v["read"]))])
        codes = cnt.code_names()
        cn = (ctypes.c_char_p * len(codes))(*[x.encode() for x in codes])
        kl = (ctypes.c_void_p * 64)(*[luts[min(i, len(luts) - 1)].ctypes.data for i in range(64)])
        bs = batch_struct(b)
        t0 = time.perf_counter()
        assert L.mmh_tie_add_batch(tie, pool, ctypes.byref(bs), rows.ctypes.data, len(rows), kl, cn, len(codes)) == 0
        t_add += time.perf_counter() - t0
        n_rows += len(rows)
    w = cnt.rows()
    out = np.zeros(len(w), dtype=ROW_DTYPE)
    for a, c in (("tid", "tid"), ("pos", "pos"), ("strand", "strand"), ("code", "code"), ("ins_offset", "ins_off"), ("hp", "hp"), ("n_called", "n_called"), ("n_mod", "n_mod")):
        out[a] = w[c]
    t0 = time.perf_counter()
    rc = L.mmh_tie_order_rows_mt(tie, pool, out.ctypes.data, len(out)); assert rc == 0 or os.environ.get("MM_TIE_SKIP")
    t_ord = time.perf_counter() - t0
    print("%d reads, %d calls, %d keys, %d threads: add_batch %.3f s (%.0f ns a call and thread), order_rows %.3f s; order md5 %s" %
          (n_reads, n_rows, len(out), threads, t_add, 1e9 * t_add * threads / max(n_rows, 1), t_ord, hashlib.md5(out.tobytes()).hexdigest()[:12]))
    L.mmh_tie_destroy(tie); L.mm_pool_destroy(pool)


if __name__ == "__main__":
    main()
