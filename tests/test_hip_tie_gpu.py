"""The device-side replay of the reference's row order (csrc/tie_kernels.hip.h, include/minimod_tie.h) against its serial restatement
(csrc/host/tieorder.c, which the CPU suite pins against the reference's goldens): the core hash table's slot order and the order
ks_introsort leaves, for key sequences of every size class -- around khash's growth bounds, with and without a put behind the last new
key, hashes of real key strings (X31 of make_key's text, src/mod.c:428-439) and random ones, few and many ties."""
import ctypes
import os

import numpy as np
import pytest

from minimod_amd import build as B

pytestmark = pytest.mark.gpu


def _libs():
    H = ctypes.CDLL(os.path.join(B.LIBDIR, "libminimod_host.so"))
    H.mmh_tie_order_plain.restype = ctypes.c_int
    H.mmh_tie_order_plain.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    D = ctypes.CDLL(B.lib_path())
    D.mm_tie_order_plain.restype = ctypes.c_int32
    D.mm_tie_order_plain.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    D.mm_tie_last_stats.argtypes = [ctypes.c_void_p]
    return H, D


def _both(h, sk, pal):
    H, D = _libs()
    n = len(h)
    h = np.ascontiguousarray(h, np.uint32)
    sk = np.ascontiguousarray(sk, np.int64)
    hs, hf, ds, df = (np.zeros(max(n, 1), np.uint32) for _ in range(4))
    assert H.mmh_tie_order_plain(h.ctypes.data, sk.ctypes.data, n, pal, hs.ctypes.data, hf.ctypes.data) == 0
    assert D.mm_tie_order_plain(0, h.ctypes.data, sk.ctypes.data, n, pal, ds.ctypes.data, df.ctypes.data) == 0
    st = np.zeros(8, np.uint64)
    D.mm_tie_last_stats(st.ctypes.data)
    return hs[:n], hf[:n], ds[:n], df[:n], st


def _x31(s):
    h = s[0]
    for c in s[1:]:
        h = (h * 31 + c) & 0xFFFFFFFF
    return h


def _upper(c):
    return int(c * 0.77 + 0.5)


BOUNDS = [_upper(4 << k) for k in range(0, 15)]   # 3, 6, 12, 25, 49, 99, ... 50463


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 12, 13, 16, 17, 18, 25, 26, 49, 99, 100, 197, 394, 1000, 1577, 1578, 2048, 2049, 2050, 5000, 12616, 12617, 50463, 70001])
def test_core_table_and_sort_at_every_size_class(n):
    rng = np.random.default_rng(n)
    h = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    sk = rng.integers(0, max(2, n // 3), n).astype(np.int64)
    for pal in (0, 1):
        hs, hf, ds, df, _ = _both(h, sk, pal)
        assert (hs == ds).all(), (n, pal, "slot order")
        assert (hf == df).all(), (n, pal, "printed order")


@pytest.mark.parametrize("seed", range(6))
def test_random_sizes_and_tie_shapes(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 400000))
    h = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    shape = seed % 3
    if shape == 0:
        sk = rng.integers(0, max(2, n // 2), n).astype(np.int64)            # pairs, as a two-code run has them
    elif shape == 1:
        sk = rng.integers(0, 7, n).astype(np.int64) << 32 | rng.integers(0, max(2, n // 8), n).astype(np.int64)   # several contigs, long runs
    else:
        sk = rng.permutation(n).astype(np.int64)                            # no ties at all
    for pal in (0, 1):
        hs, hf, ds, df, _ = _both(h, sk, pal)
        assert (hs == ds).all() and (hf == df).all(), (n, pal, shape)


def test_small_tables_with_a_busy_device():
    """k_small_epochs is ONE workgroup whose wavefronts agree through flags in LDS: with other processes' kernels on the device its wavefronts drift apart,
    and a flag read without a barrier behind the read gave wrong orders there (found by the seven fuzz campaigns running at once, one run in four; a build
    with -DMM_TIE_RACY_FLAGS has the race back).  Four child processes keep the device busy -- orderings of 300 000 keys, whose kernels fill it -- while this
    one checks 400 small ones against the host's serial replay."""
    import subprocess
    import sys
    busy = ("import sys, numpy as np\nsys.path.insert(0, %r)\nfrom tests.test_hip_tie_gpu import _both\nrng = np.random.default_rng(int(sys.argv[1]))\nfirst = True\n"
            "while True:\n    n = 300000; _both(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32), rng.integers(0, n // 3, n).astype(np.int64), 0)\n"
            "    if first: print('ready', flush=True); first = False\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    kids = [subprocess.Popen([sys.executable, "-c", busy, str(k)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for k in range(4)]
    try:
        for k in kids:
            assert k.stdout.readline().strip() == b"ready"
        rng = np.random.default_rng(77)
        for i in range(400):
            n = int(rng.integers(50, 6300))
            h = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
            sk = rng.integers(0, max(2, n // 3), n).astype(np.int64)
            hs, hf, ds, df, _ = _both(h, sk, i & 1)
            assert (hs == ds).all() and (hf == df).all(), (i, n)
    finally:
        for k in kids:
            k.kill()
            k.wait()


def test_hashes_of_real_key_strings():
    """X31 of 'chr1\\t<pos>\\t<strand>\\tm|h\\t0\\t-1' over neighbouring positions: the hashes differ in their low digits' weights only, the
    probe paths are long and crowded -- what a two-code run hands the table."""
    rng = np.random.default_rng(7)
    pos = np.sort(rng.choice(5_000_000, 60000, replace=False))
    keys = []
    for p in pos:
        st = "+-"[int(rng.integers(2))]
        for code in ("m", "h"):
            keys.append(("chr1\t%d\t%s\t%s\t0\t-1" % (p, st, code)).encode())
    n = len(keys)
    perm = np.argsort(np.arange(n) // 2 + rng.integers(0, 600, n), kind="stable")   # first met by overlapping reads, not in position order
    h = np.array([_x31(keys[i]) for i in perm], dtype=np.uint32)
    sk = np.array([int(keys[i].split(b"\t")[1]) for i in perm], dtype=np.int64)
    for pal in (0, 1):
        hs, hf, ds, df, st = _both(h, sk, pal)
        assert (hs == ds).all() and (hf == df).all()
    assert st[2] <= 4 * st[1], "a growth should settle in a few passes: %s" % st


def test_equal_hashes_and_one_site():
    """every key on one probe path (equal hashes), every row on one (contig, start): the longest chains and the longest ties"""
    n = 3000
    h = np.full(n, 0x1234567, np.uint32)
    sk = np.zeros(n, np.int64)
    hs, hf, ds, df, _ = _both(h, sk, 0)
    assert (hs == ds).all() and (hf == df).all()
    h2 = (np.arange(n) % 5).astype(np.uint32)
    hs, hf, ds, df, _ = _both(h2, sk, 1)
    assert (hs == ds).all() and (hf == df).all()


def test_three_million_keys_and_its_cost():
    """the size of a 3-Gbase two-code run (C3): exact, and the device's time for it on record"""
    rng = np.random.default_rng(99)
    n = 3_100_000
    h = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    sk = np.sort(rng.integers(0, n // 2, n)).astype(np.int64)[rng.permutation(n)]
    hs, hf, ds, df, st = _both(h, sk, 0)
    assert (hs == ds).all() and (hf == df).all()
    print("3.1 M keys: %d launches, %d growths in %d passes, %d placement rounds, %d sort levels, %d segments finished by a thread, %.1f ms on the device"
          % (st[0], st[1], st[2], st[3], st[4], st[5], st[6] / 1000.0))


# ---- the whole replay (T1 - T4) against the reference's own golden files -------------------------------------------------------------
import json
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REPLAY_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
torch.zeros(1, device="cuda")
import minimod_amd
from minimod_amd import engine as E, tie as T
from oracle import oracle as O
from oracle import pybam
from tests.cases import GOLDEN, GOLDEN_CASES

def pseudo(name):
    return dict([O.pseudo_reference(os.path.join(GOLDEN, "pseudo_%%s.npz" %% name))])

def device_replay(bam_path, contigs, c="m", m=None, insertions=False, haplotypes=False, K=512, bedmethyl=False, one_launch=False):
    mods = O.parse_mod_codes(c)
    th = O.parse_mod_threshes(m, len(mods))
    eng = hv = tie = None
    wild = [i for i, (cc, _x) in enumerate(mods) if cc == "*"]
    keep = []
    batches = [b for (_bam, b, _st) in pybam.load_batches(bam_path, K=K, B=20 * 1000 * 1000) if len(b["reads"])]
    bam = next(iter(pybam.load_batches(bam_path, K=K, B=20 * 1000 * 1000)))[0]
    if one_launch:
        batches = [pybam.concat_batches(batches)] if hasattr(pybam, "concat_batches") else batches
    ctg = [(n, l, contigs.get(n)) for n, l in zip(bam.target_name, bam.target_len)]
    mm = [(cc, x, t) for (cc, x), t in zip(mods, th)]
    eng = minimod_amd.FreqEngine(mm, ctg, insertions=insertions, haplotypes=haplotypes, stream_mode=2)
    hv = minimod_amd.FreqEngine(mm, ctg, insertions=insertions, haplotypes=haplotypes, stream_mode=2, view=2, coalesce=1)
    tie = T.TieReplay(list(bam.target_name), list(bam.target_len), insertions, haplotypes)
    for b in batches:
        eng.process(b)
        if hv.wildcard:
            hv.intern_codes_from(b)
        dev = {k: torch.from_numpy(b[k].view(np.uint8).reshape(-1).copy()).cuda() for k in ("reads", "cigar", "seq", "mm", "ml")}
        torch.cuda.synchronize()
        db = dict(reads=dev["reads"].data_ptr(), cigar=dev["cigar"].data_ptr(), seq=dev["seq"].data_ptr(), mm=dev["mm"].data_ptr(), ml=dev["ml"].data_ptr(),
                  n_reads=len(b["reads"]), n_cigar_words=len(b["cigar"]), n_seq_bytes=len(b["seq"]), n_mm_bytes=len(b["mm"]), n_ml_bytes=len(b["ml"]),
                  max_n_cigar=int(b["reads"]["n_cigar"].max()), max_l_qseq=int(b["reads"]["l_qseq"].max()))
        t = hv.submit_device(db)
        ptr, n = hv.fetch_view(t, device=True)
        codes = hv.code_names()
        luts = [E.klass_lut(th[wild[0] if wild else min(i, len(th) - 1)]) for i in range(len(codes))]
        tie.set_codes(codes, luts)
        rc = tie.add_launch(E.batch_struct(db, device=True), ptr, n)
        assert rc == 0, (rc, tie.failed())
        keep.append(dev)
    rows = eng.finalize()
    perm = tie.order_rows(rows)
    assert perm is not None, tie.failed()
    st = tie.stats()
    rows = rows[perm]
    res = np.zeros(len(rows), dtype=O.ROW_DTYPE)
    res["tid"], res["pos"], res["strand"], res["code"] = rows["tid"], rows["pos"], rows["strand"], rows["code"]
    res["ins_off"], res["hp"], res["n_called"], res["n_mod"] = rows["ins_offset"], rows["hp"], rows["n_called"], rows["n_mod"]
    text = O.format_rows(res, eng.names, eng.code_names(), insertions=insertions, haplotypes=haplotypes, bedmethyl=bedmethyl)
    eng.close(); hv.close(); tie.close()
    return text, [int(x) for x in st]

out = {}
refs = {}
for exp, bamf, ctg, kw, exact in GOLDEN_CASES:
    if exp == "test16.tsv":
        continue
    if ctg not in refs:
        refs[ctg] = pseudo(ctg)
    kw = dict(kw)
    kw.setdefault("bedmethyl", exp.endswith(".bedmethyl"))
    for K in (512, 7):
        kk = dict(kw); kk["K"] = kw.get("K", K)
        got, st = device_replay(os.path.join(GOLDEN, "data", bamf), refs[ctg], **kk)
        want = open(os.path.join(GOLDEN, "expected", exp)).read()
        out["%%s:K%%d" %% (exp, K)] = [got == want, len(got), len(want), st]
print(json.dumps(out))
'''


def test_device_replay_reproduces_the_reference_goldens_byte_for_byte():
    """every freq golden of the reference (test/test.sh:116-232) through the device-side replay: counts from a freq handle, calls from a
    view=2 handle on DEVICE-resident batches, T1 - T4 on the device, rows printed in the order it returns -- the bytes are the golden's
    (test16 aside: SURVEY section 8c), for -K 512 and -K 7 (a read's stamp must not depend on the batch it came in)"""
    r = subprocess.run([sys.executable, "-c", REPLAY_WORKER % dict(root=ROOT)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    bad = {k: v for k, v in res.items() if not v[0]}
    assert not bad, bad
    assert len(res) == 22


# ---- the row text made on the device (mm_fmt_rows) ------------------------------------------------------------------------------------
def _py_rows_text(rows, names, codes, bedmethyl, insertions, haplotypes):
    out = []
    for r in rows:
        nc, nm, pos = int(r["n_called"]), int(r["n_mod"]), int(r["pos"])
        st = "-" if r["strand"] else "+"
        if bedmethyl:
            out.append("%s\t%d\t%d\t%s\t%d\t%s\t%d\t%d\t255,0,0\t%d\t%f\n" % (names[r["tid"]], pos, pos + 1, codes[r["code"]], nc, st, pos, pos + 1, nc, float(nm) * 100 / nc))
        else:
            line = "%s\t%d\t%d\t%s\t%d\t%d\t%f\t%s" % (names[r["tid"]], pos, pos, st, nc, nm, float(nm) / nc, codes[r["code"]])
            if insertions:
                line += "\t%d" % int(r["ins_offset"])
            if haplotypes:
                line += "\t*" if r["hp"] < 0 else "\t%d" % int(r["hp"])
            out.append(line + "\n")
    return "".join(out).encode()


@pytest.mark.parametrize("fmt", [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1)], ids=["tsv", "bedmethyl", "ins", "hp", "ins_hp"])
def test_device_formatter_writes_the_reference_text(fmt):
    """random rows over the whole range of counts, positions, contig and code names: the device's bytes are those of the reference's
    fprintf formats (src/mod.c:685, :703-715; Python's %f rounds as glibc's does), row lengths included (the rows are packed)"""
    from minimod_amd import engine as E, tie as T
    bed, ins, hp = fmt
    rng = np.random.default_rng(17 + bed + 2 * ins + 4 * hp)
    n = 120000
    names = ["chr1", "chr10_KI270825v1_alt", "chrX", "c"]
    codes = ["m", "h", "76792", "hm", "a"]
    rows = np.zeros(n, E.ROW_DTYPE)
    rows["tid"] = rng.integers(0, len(names), n)
    rows["pos"] = (rng.integers(0, 1 << 31, n) >> rng.integers(0, 31, n)).astype(np.int32)
    rows["strand"] = rng.integers(0, 2, n)
    rows["ins_offset"] = (rng.integers(0, 65536, n) >> rng.integers(0, 16, n)).astype(np.uint16)
    rows["code"] = rng.integers(0, len(codes), n)
    rows["hp"] = rng.integers(-1, 5, n)
    nc = (rng.integers(1, 1 << 32, n, dtype=np.uint64) >> rng.integers(0, 31, n).astype(np.uint64)).astype(np.uint64)
    nc[nc == 0] = 1
    rows["n_called"] = nc.astype(np.uint32)
    rows["n_mod"] = (rng.random(n) * (nc + 1)).astype(np.uint64).clip(0, nc).astype(np.uint32)
    rows["n_mod"][:1000] = 0
    rows["n_mod"][1000:2000] = rows["n_called"][1000:2000]
    f = T.RowFormatter(names, codes, bedmethyl=bool(bed), insertions=bool(ins), haplotypes=bool(hp))
    got = f.format(rows)
    assert got == _py_rows_text(rows, names, codes, bed, ins, hp)
    assert f.format(rows[:1]) == _py_rows_text(rows[:1], names, codes, bed, ins, hp) and f.format(rows[:0]) == b""
    f.close()


@pytest.mark.parametrize("c,ins,hap,stays", [("m[CG]", False, False, True), ("m,h", False, False, True), ("m[*],a[*]", False, False, True), ("m[CG]", True, False, False), ("m", False, True, False)],
                         ids=["one-code", "two-codes", "star-contexts", "insertions", "haplotypes"])
def test_rows_left_on_the_device_are_the_rows_finalize_returns(c, ins, hap, stays):
    """mm_freq_finalize_device: a run whose rows all come from the dense counters leaves them in GPU memory (the CLI formats them there); with side rows
    (--insertions) or haplotype planes they are merged on the host as before.  Either way: the rows of mm_freq_finalize, and their text."""
    from minimod_amd import engine as E, tie as T
    from oracle import pybam, oracle as O
    from tests import test_hip_stream_gpu as S
    from tests.hiprun import make_engine
    rng = np.random.default_rng(4242)
    ref = S.make_ref(rng, 120000)
    recs = [S._mixed_read(rng, ref) for _ in range(90)]
    if hap:
        for r in recs:
            r.aux += b"HPC" + bytes([int(rng.integers(1, 3))])
    mods = O.parse_mod_codes(c)
    th = O.parse_mod_threshes(None, len(mods))
    eng = make_engine(mods, th, ["chrT"], [len(ref)], {"chrT": ref.encode()}, insertions=ins, haplotypes=hap)
    eng.process(pybam.flatten(recs))
    want = eng.finalize()
    assert len(want) > 500
    host, dptr, n = eng.finalize_device()
    assert n == len(want) and (dptr is not None) == stays and (host is None) == stays
    if stays:
        # (read where they lie, mm_fmt_rows_device: the whole array, a part from its middle, a host-side call in between)
        f = T.RowFormatter(["chrT"], eng.code_names(), bedmethyl=False, insertions=ins, haplotypes=hap)
        assert f.format_device(dptr, n) == f.format(want)
        assert f.format_device(dptr + 100 * E.ROW_DTYPE.itemsize, 300) == f.format(want[100:400])
        assert f.format(want[:50]) == f.format_device(dptr, 50)
        f.close()
    else:
        assert host.tobytes() == want.tobytes()
    assert eng.finalize().tobytes() == want.tobytes()   # (and the plain call afterwards still brings them over)
    eng.close()
