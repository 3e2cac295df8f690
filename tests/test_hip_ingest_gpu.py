"""Device-side BAM ingestion (include/minimod_ingest.h, csrc/ingest_kernels.hip.h, csrc/host/devloader.c) against the host loader
(csrc/host/loader.c, itself checked against the independent Python reader in tests/test_host_cpu.py): the SAME flattened batch --
read records, the four pools byte for byte, the totals load_db keeps (reference src/minimod.c:235-333) -- for every bundled BAM
and every filter flag, with group and arena sizes small enough that records straddle groups, batches close early and a group has
to be run again into a fresh arena."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BAMS = sorted(glob.glob(os.path.join(HERE, "golden", "data", "*.bam")))


def host_batch(path, **flt):
    """the whole file as ONE batch of the host loader"""
    from minimod_amd import hostlib
    got = list(hostlib.load_batches(path, K=1 << 30, B=1 << 60, threads=2, **flt))
    assert len(got) == 1
    return got[0]


def device_batches(path, sizes=None, target_bases=1 << 62, **flt):
    from minimod_amd import hostlib
    items = list(hostlib.load_batches_device(path, threads=2, sizes=sizes, target_bases=target_bases, **flt))
    stats = items[-1][1]
    return [b for b, _ in items[:-1] if b is not None], [t for _, t in items[:-1]], stats


def concat(batches):
    """several batches as the one batch the host loader makes of the same reads: pools one behind the other, offsets moved"""
    if len(batches) == 1:
        return batches[0]
    out = {k: [] for k in ("reads", "cigar", "seq", "mm", "ml")}
    o_c = o_s = o_m = o_l = 0
    for b in batches:
        r = b["reads"].copy()
        r["cigar_off"] += o_c // 4; r["seq_off"] += o_s; r["mm_off"] += o_m; r["ml_off"] += o_l
        out["reads"].append(r)
        # a batch's pool = its items (each padded to its alignment) + 64 bytes of zero slack
        for k, o in (("cigar", None), ("seq", None), ("mm", None), ("ml", None)):
            a = b[k].view(np.uint8) if k == "cigar" else b[k]
            out[k].append(a[:len(a) - 64])
        o_c += len(b["cigar"]) * 4 - 64; o_s += len(b["seq"]) - 64; o_m += len(b["mm"]) - 64; o_l += len(b["ml"]) - 64
    z = np.zeros(64, dtype=np.uint8)
    return {"reads": np.concatenate(out["reads"]), "cigar": np.concatenate(out["cigar"] + [z]).view("<u4"), "seq": np.concatenate(out["seq"] + [z]),
            "mm": np.concatenate(out["mm"] + [z]), "ml": np.concatenate(out["ml"] + [z]),
            "max_n_cigar": max(b["max_n_cigar"] for b in batches), "max_l_qseq": max(b["max_l_qseq"] for b in batches)}


def same_batch(dev, host):
    assert len(dev["reads"]) == len(host["reads"])
    for f in host["reads"].dtype.names:
        assert np.array_equal(dev["reads"][f], host["reads"][f]), f
    for k in ("cigar", "seq", "mm", "ml"):
        assert len(dev[k]) == len(host[k]), (k, len(dev[k]), len(host[k]))
        assert np.array_equal(dev[k], host[k]), k
    assert dev["max_n_cigar"] == host["max_n_cigar"] and dev["max_l_qseq"] == host["max_l_qseq"]


@pytest.mark.parametrize("path", BAMS, ids=[os.path.basename(p) for p in BAMS])
@pytest.mark.parametrize("flt", [dict(), dict(allow_secondary=True), dict(skip_supplementary=True), dict(allow_secondary=True, skip_supplementary=True)],
                         ids=["default", "secondary", "no-supp", "both"])
def test_device_batch_equals_the_host_loaders(path, flt):
    want = host_batch(path, **flt)
    got, totals, st = device_batches(path, **flt)
    assert st["err"] == 0
    if len(want["reads"]) == 0:
        assert not got
        return
    assert len(got) == 1
    same_batch(got[0], want)
    assert st["processed_reads"] == len(want["reads"]) and st["processed_bases"] == int(want["reads"]["l_qseq"].sum())


# tiny groups (8 blocks), a head room of 512 KB, arenas that hold a few hundred KB: tails in every group, batches closing on the
# arena's size, groups run again into the next arena -- the batches put end to end must still be the host loader's one batch
SMALL = dict(group_slots=3, max_blocks=8, arenas=3, max_cbytes=1 << 20, arena_bytes=(8 * 65536 + (512 << 10)) * 2, head_room=512 << 10)


@pytest.mark.parametrize("path", BAMS, ids=[os.path.basename(p) for p in BAMS])
def test_small_groups_and_arenas(path):
    want = host_batch(path)
    got, totals, st = device_batches(path, sizes=SMALL)
    assert st["err"] == 0 and (st["groups"] > 1 or os.path.getsize(path) < 600000)
    if len(want["reads"]) == 0:
        assert not got
        return
    same_batch(concat(got), want)


def test_batches_close_on_the_base_target():
    path = os.path.join(HERE, "golden", "data", "example-ont.bam")
    want = host_batch(path)
    got, totals, st = device_batches(path, sizes=dict(group_slots=4, max_blocks=4, arenas=3, max_cbytes=1 << 20, arena_bytes=64 << 20, head_room=1 << 20), target_bases=100000)
    assert len(got) > 2
    same_batch(concat(got), want)
    assert sum(t["bases"] for t in totals if t["bases"]) == int(want["reads"]["l_qseq"].sum())


def test_totals_are_the_host_loaders():
    """total / processed entries and bytes (core_t counters, src/minimod.h:190-194) of a file with secondary alignments"""
    import ctypes
    from minimod_amd import hostlib
    path = os.path.join(HERE, "golden", "data", "dna_5mCG_5hmCG_mm_with_secondary_chr22.bam")
    L = hostlib._lib()

    class loader_t(ctypes.Structure):
        _fields_ = [("bam", ctypes.c_void_p), ("allow_secondary", ctypes.c_int), ("skip_supplementary", ctypes.c_int), ("K", ctypes.c_int32), ("B", ctypes.c_int64),
                    ("last_total_reads", ctypes.c_int32), ("last_total_bytes", ctypes.c_int64), ("last_processed_bytes", ctypes.c_int64),
                    ("total_reads", ctypes.c_uint64), ("total_bytes", ctypes.c_uint64), ("processed_reads", ctypes.c_uint64), ("processed_bytes", ctypes.c_uint64),
                    ("processed_bases", ctypes.c_uint64), ("priv", ctypes.c_void_p)]
    for flt in (dict(), dict(allow_secondary=True)):
        ld = L.mmh_loader_open(path.encode(), 2, 1 << 30, 1 << 60, int(flt.get("allow_secondary", False)), 0)
        from minimod_amd.engine import mm_batch_t
        b = mm_batch_t(); more = ctypes.c_int(1)
        assert L.mmh_loader_next(ld, 0, ctypes.byref(b), ctypes.byref(more)) >= 0
        h = loader_t.from_address(ld)
        want = (h.total_reads, h.total_bytes, h.processed_reads, h.processed_bytes, h.processed_bases)
        L.mmh_loader_close(ld)
        _, _, st = device_batches(path, **flt)
        assert (st["total_reads"], st["total_bytes"], st["processed_reads"], st["processed_bytes"], st["processed_bases"]) == want


# ---- round 5: what view and -c '*' need of a device batch -- the read names (mm_ingest_arena_names) and the codes its MM tags name
# (mm_ingest_batch_codes), so that those runs take the device-side reader too
def _python_records(path, **flt):
    """the accepted records, by the independent Python reader (oracle/pybam.py)"""
    from oracle import pybam
    bam = pybam.BamFile(path)
    recs = [r for r in bam if pybam.accept(r, flt.get("allow_secondary", False), flt.get("skip_supplementary", False))]
    bam.close()
    return recs


def _walk_codes(mm_texts):
    """csrc/host/freq_main.c intern_batch_codes restated: the codes in the order a walk over the reads' MM text meets them first -- a
    group of digits is one code, a group of letters one per letter: the string from that letter on (mod.c:1146-1160)"""
    seen = []
    for s in mm_texts:
        n, p = len(s), 0
        while p < n:
            a = e = p + 2
            while e < n and s[e:e + 1] not in (b",", b";", b"?", b"."):
                e += 1
            if e > a and e - a < 16:
                code = s[a:e]
                for c in ([code] if code[:1].isdigit() else [code[m:] for m in range(len(code))]):
                    if c not in seen:
                        seen.append(c)
            while p < n and s[p:p + 1] != b";":
                p += 1
            p += 1
    return seen


@pytest.mark.parametrize("path", BAMS, ids=[os.path.basename(p) for p in BAMS])
@pytest.mark.parametrize("sizes", [None, SMALL], ids=["default", "small"])
def test_device_batches_carry_names_and_codes(path, sizes):
    from minimod_amd import hostlib
    recs = _python_records(path)
    items = list(hostlib.load_batches_device(path, threads=2, sizes=sizes, target_bases=1 << 62, names=True, codes=True))
    got = [b for b, _ in items[:-1] if b is not None]
    assert items[-1][1]["err"] == 0
    names = [n for b in got for n in b["names"]]
    assert names == [r.qname.rstrip(b"\0") for r in recs]
    at = 0
    for b in got:   # per batch: the codes of ITS reads, in the order of ITS text
        mine = recs[at:at + len(b["reads"])]
        at += len(b["reads"])
        assert b["codes"] == _walk_codes([bytes(r.mm()) for r in mine]), os.path.basename(path)
    same_batch(concat(got), host_batch(path))   # (and the names' pool changes nothing else)


def test_code_census_on_made_up_tags():
    """groups of several letters (a code per suffix), ChEBI numbers, flags, codes behind skipped groups, codes first seen late in a long text,
    a code of nine characters (the census hands the batch back: -MM_INGEST_E_CODES), in one synthetic BAM"""
    import tempfile
    from minimod_amd import hostlib, synth
    from oracle import pybam
    rng = np.random.default_rng(11)
    def rec(i, mm, n_ml):
        seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 400))
        return pybam.make_record(0, 10 + i, 0, seq, "400M", mm, [int(x) for x in rng.integers(0, 256, n_ml)], qname=("read%d" % i).encode())
    long_list = ",".join(["0"] * 700)
    texts = ["C+m?,1,2;", "C+hm,0,0;A+a.,3;", "C+21839,1;C+m,2;", "N+xyz?,0;", "C+m," + long_list + ";G+76792?,1;T+gq,0;", "C+h.;", "C-m,1;C+c?;A-17596,0;",
             "C+m?,1;" * 30 + "T+e,0;", "A+b"]
    recs = [rec(i, t, 3) for i, t in enumerate(texts * 40)]
    with tempfile.TemporaryDirectory() as d:
        bam = os.path.join(d, "c.bam")
        synth.write_bam(bam, [("chrS", 1 << 16)], [pybam.flatten(recs)], filter_fodder=False)
        items = list(hostlib.load_batches_device(bam, threads=2, names=True, codes=True))
        got = [b for b, _ in items[:-1] if b is not None]
        assert len(got) == 1 and got[0]["codes"] == _walk_codes([t.encode() for t in texts])
        assert got[0]["names"] == [r.qname.rstrip(b"\0") for r in _python_records(bam)] and len(got[0]["names"]) == len(recs)
        # a code longer than 8 characters: the caller is told to walk the text itself
        recs2 = recs[:5] + [rec(99, "C+123456789,0;", 1)]
        bam2 = os.path.join(d, "c2.bam")
        synth.write_bam(bam2, [("chrS", 1 << 16)], [pybam.flatten(recs2)], filter_fodder=False)
        items = list(hostlib.load_batches_device(bam2, threads=2, codes=True))
        assert items[0][0]["codes"] == -6
