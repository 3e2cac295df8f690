"""Random differential run of the DEVICE-side tie-order replay (csrc/tie_kernels.hip.h, include/minimod_tie.h) against the host's serial
restatement (csrc/host/tieorder.c, pinned on the reference's goldens): random mixed batches (several groups, multi-letter groups, '.' groups,
ChEBI codes, reverse reads, long insertions), random options (-c with one to four entries, `*`, --insertions, --haplotypes with HP tags), one
to four batches a run.  Both replays get the SAME calls -- the rows of a view = 2 handle -- the device's from GPU memory, the host's from a copy;
the counts come from a freq handle.  Compared: the order of the rows each prints.   usage: python tools/fuzz_tie.py <first seed> <count>"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

torch.zeros(1, device="cuda")
from minimod_amd import engine as E, hostlib, tie as TT
from oracle import pybam, oracle as O
from tests import test_hip_stream_gpu as T
from tests.hiprun import make_engine
from tests.test_tieorder_cpu import _lib


def run(seed):
    rng = np.random.default_rng(seed)
    ref = T.make_ref(rng, 120000)
    recs = [T._mixed_read(rng, ref) for _ in range(int(rng.integers(20, 110)))]
    if rng.random() < 0.6:
        for r in recs:
            if rng.random() < 0.6:
                r.aux += b"HPC" + bytes([int(rng.integers(0, 4))])
    c = ("m,h", "m[*],a[*]", "m[C],x[*]", "*", "*[C]", "m[*]", "h[CG],m[CG],a[A]", "m")[int(rng.integers(0, 8))]
    ins, hap = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
    nb = int(rng.integers(1, 5))
    cut = sorted(rng.integers(0, len(recs) + 1, size=nb - 1).tolist())
    batches = [b for b in (recs[a:b] for a, b in zip([0] + cut, cut + [len(recs)])) if b]
    mods = O.parse_mod_codes(c)
    th = O.parse_mod_threshes(None, len(mods))
    wild = [i for i, (cc, _x) in enumerate(mods) if cc == "*"]
    names, lens = ["chrT"], [len(ref)]
    eng = make_engine(mods, th, names, lens, {"chrT": ref.encode()}, insertions=ins, haplotypes=hap)
    hv = make_engine(mods, th, names, lens, {"chrT": ref.encode()}, insertions=ins, haplotypes=hap, view=2, coalesce=1)
    dt = TT.TieReplay(names, lens, ins, hap)
    L, hl = _lib()
    tn = (ctypes.c_char_p * 1)(b"chrT")
    tl = (ctypes.c_uint32 * 1)(len(ref))
    hdr = hl.mm_bam_hdr_t(1, tn, tl)
    ht = L.mmh_tie_create(ctypes.byref(hdr), int(ins), int(hap))
    pool = L.mm_pool_create(3)
    keep = []
    for b in batches:
        fb = pybam.flatten(b)
        eng.process(fb)
        if hv.wildcard:
            hv.intern_codes_from(fb)
        dev = {k: torch.from_numpy(fb[k].view(np.uint8).reshape(-1).copy()).cuda() for k in ("reads", "cigar", "seq", "mm", "ml")}
        torch.cuda.synchronize()
        db = dict(reads=dev["reads"].data_ptr(), cigar=dev["cigar"].data_ptr(), seq=dev["seq"].data_ptr(), mm=dev["mm"].data_ptr(), ml=dev["ml"].data_ptr(),
                  n_reads=len(fb["reads"]), n_cigar_words=len(fb["cigar"]), n_seq_bytes=len(fb["seq"]), n_mm_bytes=len(fb["mm"]), n_ml_bytes=len(fb["ml"]),
                  max_n_cigar=int(fb["reads"]["n_cigar"].max()), max_l_qseq=int(fb["reads"]["l_qseq"].max()))
        t = hv.submit_device(db)
        ptr, n = hv.fetch_view(t, device=True)
        codes = hv.code_names()
        luts = [E.klass_lut(th[wild[0] if wild else min(i, len(th) - 1)]) for i in range(max(len(codes), 1))]
        dt.set_codes(codes, luts[:len(codes)])
        rc = dt.add_launch(E.batch_struct(db, device=True), ptr, n)
        rows = hv.fetch_view(t)                      # the same rows, copied to the host, for the serial replay
        cn = (ctypes.c_char_p * max(1, len(codes)))(*[x.encode() for x in codes])
        hluts = [hostlib.klass_lut(th[wild[0] if wild else min(i, len(th) - 1)]) for i in range(64)]
        kl = (ctypes.c_void_p * 64)(*[x.ctypes.data for x in hluts])
        bs = E.batch_struct(fb)
        assert L.mmh_tie_add_batch(ht, pool, ctypes.byref(bs), rows.ctypes.data, len(rows), kl, cn, len(codes)) == 0
        keep.append((dev, hluts, rc))
    rows = eng.finalize()
    out = "same"
    if len(rows):
        perm = dt.order_rows(rows)
        host = np.ascontiguousarray(rows.copy())
        hrc = L.mmh_tie_order_rows_mt(ht, pool, host.ctypes.data, len(host))
        if perm is None or hrc != 0:
            out = "gave up: device %s (bits 0x%x, rc %d), host rc %d" % (perm is None, dt.failed(), getattr(dt, "last_rc", 0), hrc)
            if (perm is None) != (hrc != 0) and not (dt.failed() & 4):   # (a haplotype tag above 61 is the device's own limit)
                out = "ONE SIDE " + out
        else:
            dev_rows = rows[perm]
            if dev_rows.tobytes() != host.tobytes():
                k = [(int(a["pos"]), int(a["strand"]), int(a["code"]), int(a["ins_offset"]), int(a["hp"])) for a in dev_rows]
                h = [(int(a["pos"]), int(a["strand"]), int(a["code"]), int(a["ins_offset"]), int(a["hp"])) for a in host]
                first = next(i for i in range(len(k)) if k[i] != h[i])
                out = "ORDER DIFFERS at row %d of %d: device %s host %s" % (first, len(k), k[first:first + 3], h[first:first + 3])
    L.mmh_tie_destroy(ht); L.mm_pool_destroy(pool)
    eng.close(); hv.close(); dt.close()
    return out, c, ins, hap, nb, len(rows)


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    t0 = time.time()
    bad = gave = 0
    for seed in range(first, first + count):
        try:
            out, c, ins, hap, nb, nrows = run(seed)
        except Exception as e:   # noqa
            out, c, ins, hap, nb, nrows = "ERROR " + repr(e)[:300], "?", None, None, 0, 0
        if out.startswith("gave up"):
            gave += 1
        elif out != "same":
            bad += 1
            print("seed", seed, c, "ins", ins, "hap", hap, "batches", nb, "rows", nrows, out, flush=True)
    print("seeds %d..%d done in %.0f s, %d problems (%d runs where both replays gave up)" % (first, first + count - 1, time.time() - t0, bad, gave))
