"""The tie-order replay of the product (minimod_amd/csrc/host/tieorder.c) WITHOUT a GPU: the calls of every read come from
the oracle (view mode 2: call order, group ordinal, implicit flag), the counts from the oracle's freq mode, and the
replay must put the rows into the order the reference's own goldens have them in -- byte for byte on test5a (--insertions),
test5b (m[*]), test5c (--haplotypes), test8 and test12 (m,h), which only matched after `sort` in round 1.  This pins the
restated khash (probe sequence, growth, in-place rehash) and introsort against the reference's behaviour on CPU; the GPU
tests (tests/test_cli_gpu.py) run the same replay fed by the device."""
import ctypes
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pybam
from tests.cases import CLI_SORTED_ONLY, GOLDEN, GOLDEN_CASES

TIED = [c for c in GOLDEN_CASES if not c[4] and c[0] not in CLI_SORTED_ONLY]


def _lib():
    from minimod_amd import hostlib
    L = hostlib._lib()
    L.mmh_tie_create.restype = ctypes.c_void_p
    L.mmh_tie_create.argtypes = [ctypes.POINTER(hostlib.mm_bam_hdr_t), ctypes.c_int, ctypes.c_int]
    L.mmh_tie_add_batch.restype = ctypes.c_int
    L.mmh_tie_add_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
    L.mmh_tie_order_rows.restype = ctypes.c_int
    L.mmh_tie_order_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.mmh_tie_order_rows_mt.restype = ctypes.c_int
    L.mmh_tie_order_rows_mt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.mmh_tie_destroy.argtypes = [ctypes.c_void_p]
    L.mm_pool_create.restype = ctypes.c_void_p
    L.mm_pool_create.argtypes = [ctypes.c_int]
    L.mm_pool_destroy.argtypes = [ctypes.c_void_p]
    return L, hostlib


def replay_order(bam_path, contigs, c="m", m=None, insertions=False, haplotypes=False, K=512, threads=3, **_):
    """`minimod freq` with the oracle's counts and the PRODUCT's row-order replay; returns the output text."""
    from minimod_amd.engine import ROW_DTYPE, VIEW_ROW_DTYPE, batch_struct
    L, hostlib = _lib()
    mods = O.parse_mod_codes(c)
    th = O.parse_mod_threshes(m, len(mods))
    luts = [hostlib.klass_lut(t) for t in th]
    cnt = vw = tie = None
    pool = L.mm_pool_create(threads)
    keep = []
    for bam, batch, _st in pybam.load_batches(bam_path, K=K, B=20 * 1000 * 1000):
        if cnt is None:
            names = list(bam.target_name)
            cnt = O.Oracle(mods, th, names, insertions, haplotypes)
            vw = O.Oracle(mods, th, names, insertions, haplotypes)
            vw.set_view(2)
            for name, seq in contigs.items():
                if name in names:
                    cnt.add_contig(name, seq)
                    vw.add_contig(name, seq)
            tn = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
            tl = (ctypes.c_uint32 * len(names))(*[int(x) for x in bam.target_len])
            hdr = hostlib.mm_bam_hdr_t(len(names), tn, tl)
            keep += [tn, tl, hdr]
            tie = L.mmh_tie_create(ctypes.byref(hdr), int(insertions), int(haplotypes))
        if not len(batch["reads"]):
            continue
        cnt.process(batch, threads)
        before = len(vw.view_rows())
        vw.process(batch, threads)
        v = vw.view_rows()[before:]
        # what the device delivers for the batch: rows by read, then position (the replay puts them back into call order)
        rows = np.zeros(len(v), dtype=VIEW_ROW_DTYPE)
        first_read = batch_first[0]      # the oracle numbers reads over all batches, the device inside the batch
        rows["read"] = (v["read"] - first_read).astype(np.uint32) | (((v["prob"] >> 8) & 0xFF).astype(np.uint32) << 21)
        rows["pos"], rows["ins_offset"], rows["code"] = v["pos"], v["ins_off"], v["code"]
        rows["read_pos"] = v["read_pos"].astype(np.uint32) | (v["prob"] & 0x80000000).astype(np.uint32)
        rows["prob"] = (v["prob"] & 0xFF).astype(np.uint8)
        order = np.lexsort((rows["ins_offset"], rows["code"], rows["pos"], v["read"]))
        rows = np.ascontiguousarray(rows[order])
        batch_first[0] += len(batch["reads"])
        codes = cnt.code_names()
        cn = (ctypes.c_char_p * len(codes))(*[x.encode() for x in codes])
        wild = [i for i, (cc, _x) in enumerate(mods) if cc == "*"]
        kl = (ctypes.c_void_p * 64)(*[luts[wild[0] if wild else min(i, len(luts) - 1)].ctypes.data for i in range(64)])
        bs = batch_struct(batch)
        assert L.mmh_tie_add_batch(tie, pool, ctypes.byref(bs), rows.ctypes.data, len(rows), kl, cn, len(codes)) == 0
    want_rows = cnt.rows()
    out = np.zeros(len(want_rows), dtype=ROW_DTYPE)
    out["tid"], out["pos"], out["strand"], out["code"] = want_rows["tid"], want_rows["pos"], want_rows["strand"], want_rows["code"]
    out["ins_offset"], out["hp"], out["n_called"], out["n_mod"] = want_rows["ins_off"], want_rows["hp"], want_rows["n_called"], want_rows["n_mod"]
    assert L.mmh_tie_order_rows_mt(tie, pool, out.ctypes.data, len(out)) == 0   # (the worker pool helps where the walk is not the reference's own)
    L.mmh_tie_destroy(tie)
    L.mm_pool_destroy(pool)
    res = np.zeros(len(out), dtype=O.ROW_DTYPE)
    res["tid"], res["pos"], res["strand"], res["code"] = out["tid"], out["pos"], out["strand"], out["code"]
    res["ins_off"], res["hp"], res["n_called"], res["n_mod"] = out["ins_offset"], out["hp"], out["n_called"], out["n_mod"]
    return O.format_rows(res, cnt.names, cnt.code_names(), insertions=insertions, haplotypes=haplotypes)


batch_first = [0]


@pytest.mark.parametrize("exp,bam,ctg,kw,exact", TIED, ids=[c[0] for c in TIED])
@pytest.mark.parametrize("K", [512, 7])
def test_replayed_order_matches_reference_golden(exp, bam, ctg, kw, exact, K, request):
    contigs = request.getfixturevalue(ctg)
    batch_first[0] = 0
    got = replay_order(os.path.join(GOLDEN, "data", bam), contigs, K=K, **kw)
    want = open(os.path.join(GOLDEN, "expected", exp)).read()
    assert got == want
