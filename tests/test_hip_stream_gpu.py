"""k_stream_reads (freq_stream.hip.h): a whole read in one wavefront, the skip list, the sequence and the CIGAR merged in
sliding windows.  These cases aim at what the windows must survive -- sparse lists that run far ahead of the directory
window, lists denser than the blocks, CIGARs with more ops than the window holds, deletions / introns / insertions longer
than the 14-bit offsets of the packed words, reads whose CIGAR is shorter than the sequence -- in both orientations, with
the routing checked through the statistics pass (reads done by the kernel / handed to the tile pipeline / to the fused
kernel).  HIP vs the oracle, bit-exact; the same input through the tile pipeline alone (stream_mode 1) as a second witness."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pybam
from tests.hiprun import make_engine, to_oracle_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def make_ref(rng, n):
    return "".join(rng.choice(list("ACGT"), size=n, p=[0.2, 0.3, 0.3, 0.2]))


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def oracle_rows(recs, ref, c, th=None):
    mods = O.parse_mod_codes(c)
    o = O.Oracle(mods, O.parse_mod_threshes(th, len(mods)), ["chrT"])
    o.add_contig("chrT", ref.encode())
    o.process(pybam.flatten(recs))
    rows = o.rows()
    codes = o.code_names()
    o.close()
    return [(int(r["pos"]), "+-"[r["strand"]], int(r["n_called"]), int(r["n_mod"]), int(r["ins_off"]), int(r["hp"]),
             codes[r["code"]]) for r in rows]


def hip_rows(recs, ref, c, th=None, **kw):
    """rows and the routing statistics of one batch"""
    mods = O.parse_mod_codes(c)
    kw.setdefault("stream_mode", 3)   # these batches are far too small for the default to stream anything; 3: '.' groups too
    eng = make_engine(mods, O.parse_mod_threshes(th, len(mods)), ["chrT"], [len(ref)], {"chrT": ref.encode()}, **kw)
    eng.stats_enable(True)
    eng.process(pybam.flatten(recs))
    st = eng.stats_get()
    rows = to_oracle_rows(eng.finalize())
    codes = eng.code_names()
    eng.close()
    return [(int(r["pos"]), "+-"[r["strand"]], int(r["n_called"]), int(r["n_mod"]), int(r["ins_off"]), int(r["hp"]),
             codes[r["code"]]) for r in rows], st


def listed_read(ref, pos, cigar, seq, base, picks, rng, flag=0, code="m", extra=""):
    """A read over `seq` (BAM orientation) whose group lists the bases of kind `base` (original orientation) with the
    indices in `picks` (rising).  For a reverse read the MM counts bases of the reverse complement."""
    orig = revcomp(seq) if flag & 16 else seq
    n_b = orig.count(base)
    picks = [k for k in picks if k < n_b]
    toks, prev = [], -1
    for k in picks:
        toks.append(str(k - prev - 1))
        prev = k
    ml = [int(x) for x in rng.integers(0, 256, size=len(toks))]
    mm = "%s+%s?" % (base, code) + "".join("," + t for t in toks) + ";" + extra
    return pybam.make_record(0, pos, flag, seq, cigar, mm, ml)


def both_ways(recs, ref, c, th=None, expect_stream=None):
    want = oracle_rows(recs, ref, c, th)
    got, st = hip_rows(recs, ref, c, th)
    assert got == want
    got2, st2 = hip_rows(recs, ref, c, th, stream_mode=1)
    assert got2 == want
    assert st2["stream_done"] == 0 and st2["stream_to_tiles"] == 0
    got3, st3 = hip_rows(recs, ref, c, th, stream_mode=2)   # the lean instantiation: reads with '.' groups go through the tiles
    assert got3 == want and st3["stream_done"] + st3["stream_to_tiles"] >= st["stream_done"]
    if expect_stream is not None:
        assert st["stream_done"] == expect_stream, st
    return st


@pytest.mark.parametrize("flag", [0, 16], ids=["fwd", "rev"])
def test_sparse_and_dense_lists(flag):
    rng = np.random.default_rng(31 + flag)
    ref = make_ref(rng, 60000)
    recs = []
    # (a) three calls 8-9 kb apart: the directory window starts over between them
    seq = ref[1000:21000]
    n_c = (revcomp(seq) if flag else seq).count("C")
    recs.append(listed_read(ref, 1000, "20000M", seq, "C", [5, n_c // 2, n_c - 3], rng, flag))
    # (b) every C listed: tens of calls per block, thousands of tokens
    seq = ref[22000:34000]
    n_c = (revcomp(seq) if flag else seq).count("C")
    recs.append(listed_read(ref, 22000, "12000M", seq, "C", list(range(n_c)), rng, flag))
    # (c) every third A (the "other" class: what is not C, G, T or N), odd read length (half-filled last byte)
    seq = ref[35000:47001]
    n_a = (revcomp(seq) if flag else seq).count("A")
    recs.append(listed_read(ref, 35000, "12001M", seq, "A", list(range(0, n_a, 3)), rng, flag, code="a"))
    # (d) bursts: 70 listed in a row, a skip over thousands, 70 more, ... (rounds that end inside and outside a window)
    seq = ref[40000:60000]
    n_c = (revcomp(seq) if flag else seq).count("C")
    picks = [k for s in range(0, n_c, 1400) for k in range(s, min(s + 70, n_c))]
    recs.append(listed_read(ref, 40000, "20000M", seq, "C", picks, rng, flag))
    both_ways(recs, ref, "m[*]", expect_stream=4)
    both_ways(recs, ref, "m,a[*]", expect_stream=4)


@pytest.mark.parametrize("flag", [0, 16], ids=["fwd", "rev"])
def test_cigars_the_window_cannot_hold(flag):
    rng = np.random.default_rng(41 + flag)
    ref = make_ref(rng, 90000)
    recs = []
    # (a) thousands of tiny ops: 2M1I2M1D ... : a 64-token round spans more ops than the window has room for
    ops, seq, rp = [], [], 500
    for i in range(3000):
        seq.append(ref[rp:rp + 2]); ops.append("2M"); rp += 2
        if i % 2 == 0:
            seq.append("C"); ops.append("1I")
        else:
            ops.append("1D"); rp += 1
    seq.append(ref[rp:rp + 5]); ops.append("5M")
    seq = "".join(seq)
    n_c = (revcomp(seq) if flag else seq).count("C")
    recs.append(listed_read(ref, 500, "".join(ops), seq, "C", list(range(0, n_c, 2)), rng, flag))
    # (b) a 20 kb deletion, (c) a 30 kb intron, (d) an 18 kb insertion between densely listed stretches
    for cig, seq in (("3000M20000D3000M", ref[100:3100] + ref[23100:26100]),
                     ("3000M30000N3000M", ref[100:3100] + ref[33100:36100]),
                     ("500M18000I500M", ref[1000:1500] + make_ref(rng, 18000) + ref[1500:2000])):
        n_c = (revcomp(seq) if flag else seq).count("C")
        recs.append(listed_read(ref, 100 if cig[0] == "3" else 1000, cig, seq, "C", list(range(n_c)), rng, flag))
    # (e) soft clips at both ends carrying listed bases, a match of 17 kb (an op longer than the packed offsets), a CIGAR
    #     that ends before the sequence does (the bases behind it have no call)
    seq = make_ref(rng, 700) + ref[50000:67000] + make_ref(rng, 900)
    n_c = (revcomp(seq) if flag else seq).count("C")
    recs.append(listed_read(ref, 50000, "700S17000M900S", seq, "C", list(range(0, n_c, 5)), rng, flag))
    seq = ref[70000:76000]
    n_c = (revcomp(seq) if flag else seq).count("C")
    recs.append(listed_read(ref, 70000, "100S5000M", seq, "C", list(range(0, n_c, 3)), rng, flag))
    both_ways(recs, ref, "m[*]", expect_stream=6)
    both_ways(recs, ref, "m", expect_stream=6)


def test_groups_in_sequence_and_what_is_handed_on():
    rng = np.random.default_rng(51)
    ref = make_ref(rng, 30000)
    seq = ref[2000:14000]
    n_c, n_g = seq.count("C"), seq.count("G")
    rc = revcomp(seq)
    recs = []
    # two requested groups on one base, an unrequested one between them (its tokens still move the ML index)
    def three(flag):
        s = rc if flag else seq
        nc = s.count("C")
        t1, t2, t3 = list(range(0, nc, 4)), list(range(1, nc, 7)), list(range(2, nc, 5))
        def toks(p):
            out, prev = [], -1
            for k in p:
                out.append(str(k - prev - 1)); prev = k
            return out
        mm = "C+h?" + "".join("," + t for t in toks(t1)) + ";C+x?" + "".join("," + t for t in toks(t2)) + ";C+m?" + "".join("," + t for t in toks(t3)) + ";"
        ml = [int(x) for x in rng.integers(0, 256, size=len(t1) + len(t2) + len(t3))]
        return pybam.make_record(0, 2000, flag, seq, "12000M", mm, ml)
    recs += [three(0), three(16)]
    # a two-letter group: two ML bytes per token
    t = list(range(0, n_c, 6))
    prev, tk = -1, []
    for k in t:
        tk.append(str(k - prev - 1)); prev = k
    recs.append(pybam.make_record(0, 2000, 0, seq, "12000M", "C+hm?" + "".join("," + x for x in tk) + ";", [int(x) for x in rng.integers(0, 256, size=2 * len(tk))]))
    st = both_ways(recs, ref, "m,h", expect_stream=3)
    assert st["stream_to_tiles"] == 0 and st["stream_to_fused"] == 0
    st = both_ways(recs, ref, "m[CG]", expect_stream=3)
    # not this kernel's reads: a group on N, groups on two bases -> the tile pipeline (or its fallback); same rows.  (A '.' group
    # is this kernel's: its unlisted bases are calls too.)
    other = [pybam.make_record(0, 2000, 0, seq, "12000M", "C+m.,3,5,7;", [200, 10, 255]),
             pybam.make_record(0, 2000, 0, seq, "12000M", "N+m?,30,50,70;", [200, 10, 255]),
             pybam.make_record(0, 2000, 0, seq, "12000M", "C+m?,3,5,7;G+m?,1,1;", [200, 10, 255, 3, 250]),
             pybam.make_record(0, 2000, 0, seq, "12000M", "", [])]
    st = both_ways(recs + other, ref, "m", expect_stream=5)   # the read with an empty MM is done by the stream kernel (nothing to do)
    assert st["stream_to_tiles"] == 2 and st["stream_to_fused"] == 0


def test_errors_after_the_first_counts_go_to_the_fused_kernel():
    """A malformed token, a rank past the read's last C and an ML array that is too short, each far enough into the list
    that the stream kernel has counted calls before it meets them: the fused kernel names the error as the reference does."""
    import minimod_amd
    rng = np.random.default_rng(61)
    ref = make_ref(rng, 12000)
    seq = ref[1000:9000]
    n_c = seq.count("C")
    good = pybam.make_record(0, 1000, 0, seq, "8000M", "C+m?,0,1;", [255, 0])
    cases = []
    toks = ["0"] * 300
    toks[250] = "1x"
    cases.append(("C+m?" + "".join("," + t for t in toks) + ";", [200] * 300))
    toks = ["0"] * 200 + [str(n_c)]
    cases.append(("C+m?" + "".join("," + t for t in toks) + ";", [200] * 201))
    toks = ["0"] * 400
    cases.append(("C+m?" + "".join("," + t for t in toks) + ";", [200] * 350))
    for mm, ml in cases:
        rec = pybam.make_record(0, 1000, 0, seq, "8000M", mm, ml)
        o = O.Oracle(O.parse_mod_codes("m[*]"), [0.8], ["chrT"])
        o.add_contig("chrT", ref.encode())
        with pytest.raises(O.OracleError) as oe:
            o.process(pybam.flatten([good, rec, good]))
        o.close()
        eng = make_engine(O.parse_mod_codes("m[*]"), [0.8], ["chrT"], [len(ref)], {"chrT": ref.encode()}, stream_mode=3)
        with pytest.raises(minimod_amd.MinimodHipError) as he:
            eng.process(pybam.flatten([good, rec, good]))
        eng.close()
        assert (he.value.code, he.value.read) == (oe.value.code, 1)


def test_synthetic_reads_are_streamed():
    """ONT- and HiFi-shape synthetic reads (the bench's generators): every read of at most split_bases bases is done by the
    stream kernel ('.' reads with their implicit calls included), longer ones by the tile pipeline; rows equal the oracle's."""
    import minimod_amd
    from minimod_amd import synth
    ref = synth.reference(23, 4 << 20)
    for gen, mods, th in ((dict(n=500, max_len=0.0, dot_fraction=0.2), [("m", "CG")], [0.8]),
                          (dict(n=400, shape=1), [("m", "CG"), ("h", "CG")], [0.8, 0.7]),
                          (dict(n=300, max_len=0.0), [("h", "CG")], [0.6])):
        g = dict(gen); n = g.pop("n")
        b = synth.batch(ref, 0, n, seed=91, n_reads_total=n, **g)
        eng = minimod_amd.FreqEngine([(c, x, t) for (c, x), t in zip(mods, th)], [("chrS", len(ref), ref)], stream_mode=3)
        eng.stats_enable(True)
        eng.process(b)
        st = eng.stats_get()
        got = eng.finalize(); eng.close()
        orc = O.Oracle(mods, th, ["chrS"]); orc.add_contig("chrS", ref); orc.process(b, threads=8)
        want = orc.rows()
        key = lambda r, io: sorted(zip(r["pos"].tolist(), r["strand"].tolist(), r["code"].tolist(), r[io].tolist(), r["n_called"].tolist(), r["n_mod"].tolist()))
        assert key(got, "ins_offset") == key(want, "ins_off")
        rd = b["reads"]
        dots = np.array([bytes(b["mm"][int(o):int(o) + 4]).endswith(b".") for o in rd["mm_off"]])
        short = rd["l_qseq"] <= 24576
        assert dots.any() or g.get("dot_fraction", 0) == 0
        assert st["stream_done"] == int(short.sum()), st          # '?' and '.' reads alike
        assert st["stream_to_tiles"] == 0 and st["stream_to_fused"] == 0, st


def _random_read(rng, ref, flag):
    """a read with a random CIGAR (M = X I D N, soft clips at the ends, now and then a very long D / N / I) and one to three
    groups on C with random skip lists"""
    n_ops = int(rng.integers(1, 60))
    pos = int(rng.integers(0, 2000))
    rp, ops, seq = pos, [], []
    if rng.random() < 0.4:
        l = int(rng.integers(1, 300)); ops.append("%dS" % l); seq.append(make_ref(rng, l))
    for i in range(n_ops):
        l = int(rng.geometric(0.02)) if rng.random() < 0.8 else int(rng.integers(200, 3000))
        if rp + l >= len(ref) - 40000:
            break
        kind = "M" if i == 0 or i == n_ops - 1 else str(rng.choice(list("MMMM=XIDN")))
        if kind in "M=X":
            seq.append(ref[rp:rp + l]); rp += l
        elif kind == "I":
            l = l if rng.random() < 0.95 else int(rng.integers(16000, 20000))
            seq.append(make_ref(rng, l))
        else:
            l = l if rng.random() < 0.9 else int(rng.integers(16000, 35000))
            rp += l
        ops.append("%d%s" % (l, kind))
    if ops[-1][-1] in "IDN":
        ops.append("7M"); seq.append(ref[rp:rp + 7]); rp += 7
    if rng.random() < 0.4:
        l = int(rng.integers(1, 300)); ops.append("%dS" % l); seq.append(make_ref(rng, l))
    seq = "".join(seq)
    orig = revcomp(seq) if flag else seq
    n_c = orig.count("C")
    mm, ml = "", []
    for code in rng.permutation(["m", "h", "x"])[:int(rng.integers(1, 4))]:
        dens = float(rng.choice([0.02, 0.2, 0.7, 1.0]))
        picks = [k for k in range(n_c) if rng.random() < dens]
        if rng.random() < 0.15:
            picks = picks[:3]
        toks, prev = [], -1
        for k in picks:
            toks.append(str(k - prev - 1).zfill(int(rng.integers(1, 4)) if rng.random() < 0.1 else 1)); prev = k
        mm += "C+%s?" % code + "".join("," + t for t in toks) + ";"
        ml += [int(x) for x in rng.integers(0, 256, size=len(toks))]
    return pybam.make_record(0, pos, flag, seq, "".join(ops), mm, ml)


@pytest.mark.parametrize("seed", range(6))
def test_random_reads_against_the_oracle(seed):
    rng = np.random.default_rng(700 + seed)
    ref = make_ref(rng, 400000)
    recs = [_random_read(rng, ref, 16 if rng.random() < 0.5 else 0) for _ in range(120)]
    for c in ("m,h", "m[*],h[*]", "h[CG]"):
        st = both_ways(recs, ref, c)
        short = sum(1 for r in recs if r.l_qseq <= 24576)   # (a read with one of the long insertions can be longer: the tile pipeline's)
        assert st["stream_done"] == short and st["stream_to_tiles"] == 0 and st["stream_to_fused"] == 0, st


@pytest.mark.parametrize("config", ["C2", "C3", "C5"])
def test_full_size_timed_launch_equals_the_oracle(config, tmp_path):
    """BASELINE.json's workloads at full size, through exactly what bench.py times: C2 = 100 000 ONT-shape reads as 25
    device-resident -K 4096 windows gathered into one launch (every read goes through k_stream_reads there, the 100 kb ones
    too); C3 = 100 000 HiFi-shape reads, -c m[CG],h[CG] -m 0.8,0.7 (twin groups); C5 = 66 000 reads at 200x on 5 Mb with
    --haplotypes --insertions (the kIns instantiation, 50 million side rows).  The rows the timed engine holds after one pass
    equal the oracle's over the same batches, row for row."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    wl = bench.WORKLOADS[config]
    reads = wl.get("reads", 100000)
    nb = (reads + 4095) // 4096
    dump = str(tmp_path / "timed.npz")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", config, "--steps", str(nb), "--warmup", "0", "--reps", "2", "--no-e2e",
                        "--no-cpu-baseline", "--no-extra", "--dump-timed", dump], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["dump_timed"]["batches"] == nb and d["roofline"]["launches"] == 1
    rt = d["config"]["routing"]
    assert rt["reads"] == reads and rt["handed_to_the_fused_kernel"] == 0 and rt["reads_done_by_k_stream_reads"] >= 0.99 * reads
    got = np.load(dump)["rows"]
    region = wl.get("region", bench.INTERVAL)
    plan = bench.shard_plan(0, 1, region)
    ref = bench.gen_reference(plan, 0x5EED)
    orc = O.Oracle([(c, x) for c, x, _ in wl["mods"]], [t for _, _, t in wl["mods"]], ["chrS"], **wl["eng"])
    orc.add_contig("chrS", ref)
    for bi in range(nb):
        orc.process(bench.gen_batch(ref, plan, 0, 0x5EED, reads, 4096, bi, **wl["gen"]), threads=os.cpu_count() or 1)
    want = orc.rows()
    assert len(want) > 500000 and len(got) == len(want)
    for a, b in (("pos", "pos"), ("strand", "strand"), ("code", "code"), ("ins_offset", "ins_off"), ("hp", "hp"), ("n_called", "n_called"), ("n_mod", "n_mod")):
        assert (got[a] == want[b]).all(), a


@pytest.mark.parametrize("flag", [0, 16], ids=["fwd", "rev"])
def test_dot_groups_and_their_implicit_calls(flag):
    """'.' groups: every base of the class the list skips is a call too (called, not modified), and so are the bases behind the
    last listed one.  Dense lists (no gaps at all), sparse ones (gaps of thousands: far more implicit calls than tokens), an
    empty list (the whole read implicit), a two-letter group, a '.' group next to a '?' group and to one nobody asked for."""
    rng = np.random.default_rng(81 + flag)
    ref = make_ref(rng, 50000)
    recs = []
    def dot(mm):
        return mm.replace("?", ".", 1)
    seq = ref[1000:9000]
    n_c = (revcomp(seq) if flag else seq).count("C")
    for picks in (list(range(n_c)), list(range(0, n_c, 2)), [7, n_c // 3, n_c // 3 + 1, n_c - 400], list(range(5, n_c, 97)), []):
        r = listed_read(ref, 1000, "8000M", seq, "C", picks, rng, flag)
        mm = r.mm().decode() if isinstance(r.mm(), bytes) else r.mm()
        recs.append(pybam.make_record(0, 1000, flag, seq, "8000M", dot(mm), list(r.ml() or b"")))
    # soft clips and an insertion / deletion in the way of the implicit calls, a CIGAR shorter than the sequence
    seq2 = make_ref(rng, 300) + ref[20000:23000] + make_ref(rng, 40) + ref[23500:26000] + make_ref(rng, 200)
    n_c2 = (revcomp(seq2) if flag else seq2).count("C")
    r = listed_read(ref, 20000, "300S3000M40I500D2500M200S", seq2, "C", list(range(0, n_c2, 5)), rng, flag)
    recs.append(pybam.make_record(0, 20000, flag, seq2, "300S3000M40I500D2500M200S", dot(r.mm().decode() if isinstance(r.mm(), bytes) else r.mm()), list(r.ml() or b"")))
    seq3 = ref[30000:34000]
    n_c3 = (revcomp(seq3) if flag else seq3).count("C")
    r = listed_read(ref, 30000, "3500M", seq3, "C", list(range(0, n_c3, 3)), rng, flag)
    recs.append(pybam.make_record(0, 30000, flag, seq3, "3500M", dot(r.mm().decode() if isinstance(r.mm(), bytes) else r.mm()), list(r.ml() or b"")))
    # a two-letter '.' group, then a '?' group, then a '.' group with a code nobody asked for
    s4 = ref[40000:46000]
    o4 = revcomp(s4) if flag else s4
    nc4 = o4.count("C")
    def toks(p):
        out, prev = [], -1
        for k in p:
            out.append(str(k - prev - 1)); prev = k
        return out
    t1, t2, t3 = list(range(0, nc4, 6)), list(range(1, nc4, 9)), list(range(2, nc4, 11))
    mm = "C+hm." + "".join("," + t for t in toks(t1)) + ";C+m?" + "".join("," + t for t in toks(t2)) + ";C+x." + "".join("," + t for t in toks(t3)) + ";"
    ml = [int(x) for x in rng.integers(0, 256, size=2 * len(t1) + len(t2) + len(t3))]
    recs.append(pybam.make_record(0, 40000, flag, s4, "6000M", mm, ml))
    for c in ("m[*]", "m", "m,h", "h[CG]"):
        st = both_ways(recs, ref, c, expect_stream=len(recs))
        assert st["stream_to_tiles"] == 0 and st["stream_to_fused"] == 0, st


def test_the_kernel_switches_to_its_dot_instantiation_by_itself():
    """stream_mode 2 (and the default): the lean instantiation runs until a read with a '.' group shows up -- that launch's '.'
    reads go through the tile pipeline -- and the '.'-capable one from the next launch on.  Same rows either way."""
    import minimod_amd
    from minimod_amd import synth
    ref = synth.reference(29, 4 << 20)
    b = synth.batch(ref, 0, 400, seed=93, n_reads_total=400, max_len=20000.0, dot_fraction=0.5)
    dots = int(np.array([bytes(b["mm"][int(o):int(o) + 4]).endswith(b".") for o in b["reads"]["mm_off"]]).sum())
    assert 100 < dots < 300
    eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)], stream_mode=2)
    eng.stats_enable(True)
    eng.process(b)
    st1 = eng.stats_get()
    eng.process(b)
    st2 = eng.stats_get()
    eng.process(b)
    st3 = eng.stats_get()
    got = eng.finalize(); eng.close()
    assert st1["stream_to_tiles"] == dots and st1["stream_done"] == 400 - dots, st1
    assert st3["stream_to_tiles"] == 0 and st3["stream_done"] == 400, (st2, st3)
    orc = O.Oracle([("m", "CG")], [0.8], ["chrS"]); orc.add_contig("chrS", ref)
    for _ in range(3):
        orc.process(b, threads=8)
    want = orc.rows()
    key = lambda r, io: sorted(zip(r["pos"].tolist(), r["strand"].tolist(), r["n_called"].tolist(), r["n_mod"].tolist()))
    assert key(got, None) == key(want, None)


WAIT_TIME_WORKER = r'''
import json, sys
sys.path.insert(0, %r)
import os

import numpy as np
import torch
torch.zeros(1, device="cuda")
import minimod_amd
from minimod_amd import synth
from oracle import oracle as O
ref = synth.reference(37, 4 << 20)
b = synth.batch(ref, 0, 500, seed=95, n_reads_total=500, max_len=0.0, dot_fraction=0.0, with_order=False)
dev = {k: torch.from_numpy(b[k].view(np.uint8).reshape(-1)).cuda() for k in ("reads", "cigar", "seq", "mm", "ml")}
win = dict(reads=dev["reads"].data_ptr(), cigar=dev["cigar"].data_ptr(), seq=dev["seq"].data_ptr(), mm=dev["mm"].data_ptr(), ml=dev["ml"].data_ptr(),
           n_reads=500, n_cigar_words=len(b["cigar"]), n_seq_bytes=len(b["seq"]), n_mm_bytes=len(b["mm"]), n_ml_bytes=len(b["ml"]),
           max_n_cigar=int(b["max_n_cigar"]), max_l_qseq=100)
eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)], stream_mode=2)
eng.stats_enable(True)
eng.wait(eng.submit_device(win))
st = eng.stats_get()
got = eng.finalize(); eng.close()
orc = O.Oracle([("m", "CG")], [0.8], ["chrS"]); orc.add_contig("chrS", ref); orc.process(b, threads=8)
want = orc.rows()
key = lambda r: sorted(zip(r["pos"].tolist(), r["strand"].tolist(), r["n_called"].tolist(), r["n_mod"].tolist()))
print(json.dumps({"equal": key(got) == key(want), "rows": int(len(want)), "stream_done": int(st["stream_done"]),
                  "long_reads": int((b["reads"]["l_qseq"] > 24576).sum()), "max_l": int(b["reads"]["l_qseq"].max())}))
'''


def test_tile_kernels_run_at_wait_time_when_the_launch_left_them_out():
    """When the batch's longest read is a stream item by the host's reckoning the tile kernels are not launched with the
    batch; they run at wait time if k_stream_reads handed a read on, or if the planner found tile items after all -- here
    because the caller's max_l_qseq is wrong (reads of 40 kb and more in a batch that says 100)."""
    import subprocess, sys, json
    r = subprocess.run([sys.executable, "-c", WAIT_TIME_WORKER % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert res["max_l"] > 40000 and res["long_reads"] > 10, res
    assert res["equal"] and res["rows"] > 1000 and res["stream_done"] == 500 - res["long_reads"], res


def test_reference_letters_other_than_acgt():
    """One requested mod keeps the reference at four bits a position (RefNib: base in two bits, any other letter stored as
    A): N, IUPAC letters and lower case in the reference, N and IUPAC letters in the reads, contexts of one to three letters,
    `*`, and two with an N in them (the option parser takes A C G T U N; such a run keeps 16-bit words).  Stream kernel, tile pipeline and fused kernel
    against the oracle."""
    rng = np.random.default_rng(4242)
    clean = make_ref(rng, 120000)
    ref = list(clean)
    for i in rng.choice(len(ref), size=len(ref) // 25, replace=False):
        ref[i] = str(rng.choice(list("NNNRYMKSWnacgt")))
        if ref[i] in "acgt":
            ref[i] = clean[i].lower()
    ref = "".join(ref)
    recs = []
    for _ in range(80):
        flag = 16 if rng.random() < 0.5 else 0
        pos = int(rng.integers(0, 60000)); l = int(rng.integers(200, 9000))
        seq = list(clean[pos:pos + l])
        for i in rng.choice(l, size=l // 40, replace=False):
            seq[i] = str(rng.choice(list("NRYACGT")))
        seq = "".join(seq)
        orig = "".join({"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "R": "Y", "Y": "R"}[c] for c in reversed(seq)) if flag else seq
        n_c = orig.count("C")
        picks = [k for k in range(n_c) if rng.random() < 0.6]
        toks, prev = [], -1
        for k in picks:
            toks.append(str(k - prev - 1)); prev = k
        mm = "C+m?" + "".join("," + t for t in toks) + ";"
        recs.append(pybam.make_record(0, pos, flag, seq, "%dM" % l, mm, [int(x) for x in rng.integers(0, 256, size=len(toks))]))
    for c in ("m[CG]", "m[*]", "m[C]", "m[CGN]", "m[NG]", "m[CCG]"):
        want = oracle_rows(recs, ref, c)
        for kw in (dict(stream_mode=2), dict(stream_mode=1), dict(force_fused=True)):
            got, _ = hip_rows(recs, ref, c, **kw)
            assert got == want, (c, kw)
        assert len(want) > 0 or c in ("m[CGN]", "m[NG]")


COMP_N = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def _mixed_read(rng, ref):
    """a read of a mixed batch: one to four groups with `?`, `.` or no flag, mostly on one base but sometimes on several
    (such a read is the fused kernel's: it is irregular from its headers on), any of A C G T N, one or two codes a group,
    lists from empty to every base, zero-padded tokens, mismatches and N in the sequence"""
    flag = 16 if rng.random() < 0.5 else 0
    n_ops = int(rng.integers(1, 40))
    pos = int(rng.integers(0, 3000))
    rp, ops, seq = pos, [], []
    if rng.random() < 0.3:
        l = int(rng.integers(1, 100)); ops.append("%dS" % l); seq.append(make_ref(rng, l))
    for i in range(n_ops):
        l = int(rng.geometric(0.03)) if rng.random() < 0.85 else int(rng.integers(100, 1500))
        kind = "M" if i == 0 or i == n_ops - 1 else str(rng.choice(list("MMMM=XIDN")))
        if kind in "M=X":
            s = list(ref[rp:rp + l])
            for j in range(len(s)):
                if rng.random() < 0.03:
                    s[j] = str(rng.choice(list("ACGTN")))
            seq.append("".join(s)); rp += l
        elif kind == "I":
            seq.append(make_ref(rng, l))
        else:
            rp += l
        ops.append("%d%s" % (l, kind))
    if ops[-1][-1] in "IDN":
        ops.append("5M"); seq.append(ref[rp:rp + 5]); rp += 5
    seq = "".join(seq)
    orig = "".join(COMP_N[c] for c in reversed(seq)) if flag else seq
    mm, ml = "", []
    same_base = rng.random() < 0.8
    base0 = str(rng.choice(list("CCCCCAGTN")))
    for g in range(int(rng.integers(1, 5))):
        base = base0 if same_base else str(rng.choice(list("CCCAGTN")))
        n_b = len(orig) if base == "N" else orig.count(base)
        codes = "".join(rng.permutation(list("mhxa"))[:int(rng.integers(1, 3))])
        fl = str(rng.choice(["?", "?", ".", ""]))
        dens = float(rng.choice([0.0, 0.02, 0.3, 0.9, 1.0]))
        picks = [k for k in range(n_b) if rng.random() < dens]
        toks, prev = [], -1
        for k in picks:
            toks.append(str(k - prev - 1).zfill(int(rng.integers(1, 4)) if rng.random() < 0.05 else 1)); prev = k
        mm += "%s+%s%s" % (base, codes, fl) + "".join("," + t for t in toks) + ";"
        ml += [int(x) for x in rng.integers(0, 256, size=len(toks) * len(codes))]
    return pybam.make_record(0, pos, flag, seq, "".join(ops), mm, ml)


@pytest.mark.parametrize("seed", [197, 202, 206, 213, 216, 217, 1001, 1002])
def test_mixed_batches_against_the_oracle(seed):
    """Batches in which regular reads, reads for the fused kernel (groups on different bases) and reads with implicit calls
    share the tile regions.  Found by this generator: a read that was irregular from its headers on (so never reserved tile
    records) marked the first records of its region invalid -- another read's, whose calls were then lost or not, depending
    on which wavefront came first.  Every mode, three times over (the loss was a race)."""
    rng = np.random.default_rng(seed)
    ref = make_ref(rng, 120000)
    recs = [_mixed_read(rng, ref) for _ in range(int(rng.integers(60, 120)))]
    for c in ("m", "m,h", "m[*],a[*]"):
        want = oracle_rows(recs, ref, c)
        for kw in (dict(stream_mode=1), dict(stream_mode=2), dict(stream_mode=3), dict(force_fused=True)):
            for _ in range(3 if "force_fused" not in kw else 1):
                got, _st = hip_rows(recs, ref, c, **kw)
                assert got == want, (seed, c, kw)


def _twin_read(rng, ref, pos, length, flag, variant):
    """a read whose two groups list the same bases (as 5mC + 5hmC callers write them), or nearly so"""
    seq = ref[pos:pos + length]
    orig = revcomp(seq) if flag else seq
    n_c = orig.count("C")
    picks = [k for k in range(n_c) if rng.random() < 0.4]
    toks, prev = [], -1
    for k in picks:
        toks.append(str(k - prev - 1)); prev = k
    la = "".join("," + t for t in toks)
    lb = la
    if variant == "padded":           # zero-padded tokens in both lists: still plain text
        la = lb = "".join("," + t.zfill(int(rng.integers(1, 4))) for t in toks)
    elif variant == "differs":        # the same length, two neighbouring tokens the other way round: two passes
        tb = list(toks)
        for i in range(len(tb) - 1):
            if tb[i] != tb[i + 1]:
                tb[i], tb[i + 1] = tb[i + 1], tb[i]
                break
        lb = "".join("," + t for t in tb)
    elif variant == "shorter":        # the second list stops early
        lb = "".join("," + t for t in toks[:len(toks) // 2])
    elif variant == "empty_token":    # ",," is not plain: the groups go one after the other
        la = lb = la.replace(",", ",,", 1)
    n_a, n_b = la.count(",") - la.count(",,"), lb.count(",") - lb.count(",,")
    third = "C+x?" + "".join("," + t for t in toks[:5]) + ";" if variant == "third" else ""
    n_x = min(5, len(toks)) if variant == "third" else 0
    mm = "C+h?" + la + ";C+m?" + lb + ";" + third
    ml = [int(x) for x in rng.integers(0, 256, size=n_a + n_b + n_x)]
    if variant == "ml_short":
        ml = ml[:n_a + n_b // 2]
    return pybam.make_record(0, pos, flag, seq, "%dM" % length, mm, ml)


@pytest.mark.parametrize("variant", ["same", "padded", "differs", "shorter", "empty_token", "third"])
def test_twin_groups(variant):
    """Two `?` groups of one requested code each over the same list are done in one pass (RefWord 16 / 32 bit: two requested
    codes); lists that only look alike are not.  Rows, and the view rows' order, against the oracle; the tile pipeline as a
    second witness."""
    rng = np.random.default_rng(77)
    ref = make_ref(rng, 60000)
    recs = [_twin_read(rng, ref, int(rng.integers(0, 30000)), int(rng.integers(300, 9000)), 16 if i % 2 else 0, variant) for i in range(40)]
    for c in ("m,h", "h[C],m[CG]", "m[*],h[*],x[*]"):
        st = both_ways(recs, ref, c)
        assert st["stream_done"] == len(recs), st
    # one requested code of the pair: nothing to pair (and with one mod the reference array is four bits a base)
    both_ways(recs, ref, "m")
    from tests.hiprun import hip_view_from_records
    mods = O.parse_mod_codes("m,h")
    o = O.Oracle(mods, O.parse_mod_threshes(None, 2), ["chrT"]); o.set_view(True); o.add_contig("chrT", ref.encode()); o.process(pybam.flatten(recs))
    codes = o.code_names()
    want = [(int(r["read"]), int(r["pos"]), int(r["read_pos"]), codes[r["code"]], int(r["prob"]), int(r["ins_off"])) for r in o.view_rows()]
    o.close()
    for mode in (2, 1):
        got = hip_view_from_records(recs, ref, "m,h", stream_mode=mode)
        assert got == want, mode


def test_twin_groups_with_too_few_ml_bytes():
    """the second list's ML bytes run out: the reference's error (ML index overrun) at the same read"""
    import minimod_amd
    rng = np.random.default_rng(78)
    ref = make_ref(rng, 60000)
    recs = [_twin_read(rng, ref, 100 + 50 * i, 2000, 0, "same") for i in range(6)]
    recs[3] = _twin_read(rng, ref, 700, 2500, 0, "ml_short")
    mods = O.parse_mod_codes("m,h")
    o = O.Oracle(mods, O.parse_mod_threshes(None, 2), ["chrT"]); o.add_contig("chrT", ref.encode())
    with pytest.raises(O.OracleError) as oe:
        o.process(pybam.flatten(recs))
    o.close()
    assert (oe.value.code, oe.value.read) == (11, 3)
    for mode in (2, 3, 1):
        with pytest.raises(minimod_amd.MinimodHipError) as he:
            hip_rows(recs, ref, "m,h", stream_mode=mode)
        assert (he.value.code, he.value.read) == (11, 3), mode


# ---- ABI 6: up to 32 -c entries; entries with one context string share its bits of the reference word (13 DIFFERENT contexts at most)
def _entries(codes, contexts):
    return ",".join("%s[%s]" % (c, contexts[i % len(contexts)]) for i, c in enumerate(codes))


MANY_CODES = ["m", "h", "x", "a", "21839", "76792", "b", "c", "d", "e", "f", "g", "i", "j", "k", "l", "n", "o", "p", "q", "r", "s", "t", "u", "v", "w", "y", "z", "17802", "19228", "27301", "1"]


@pytest.mark.parametrize("n_entries,contexts", [(14, ["CG", "A", "*"]), (20, ["CG", "A", "*", "C", "CA"]), (32, ["CG", "A", "*", "C", "CA", "CT", "T", "G"]),
                                                (16, ["CG"]), (32, ["*"]), (13, ["CG", "A", "*", "C", "CT", "CC", "T", "G", "AC", "GC", "TA", "CA", "GG"])],
                         ids=["14x3", "20x5", "32x8", "16x1", "32xstar", "13x13"])
def test_more_than_thirteen_entries(n_entries, contexts):
    """The reference takes any number of -c entries (src/minimod.h:114); rounds 1 - 4 took 13 -- two context bits an entry in a 32-bit reference
    word.  Since ABI 6 the bits belong to the context STRING: 32 entries, 13 different contexts.  Mixed reads (groups on every base, one or
    two codes a group, all three flags) and regular ones, through the stream kernel, the tile pipeline and the fused kernel, against the oracle --
    with three, five (16-bit words), eight (32-bit words) and thirteen contexts"""
    rng = np.random.default_rng(4100 + n_entries)
    ref = make_ref(rng, 200000)
    recs = [_random_read(rng, ref, 16 if rng.random() < 0.5 else 0) for _ in range(40)] + [_mixed_read(rng, ref) for _ in range(80)]
    c = _entries(MANY_CODES[:n_entries], contexts)
    both_ways(recs, ref, c)
    # a permutation of the entries: the same rows (code indices follow the entries, the context bits follow the contexts)
    perm = list(rng.permutation(n_entries))
    c2 = ",".join(c.split(",")[i] for i in perm)
    assert sorted(oracle_rows(recs, ref, c2)) == sorted(hip_rows(recs, ref, c2)[0])


MANY_CONTEXTS = ["CG", "A", "C", "CT", "CC", "T", "G", "AC", "GC", "TA", "CA", "GG", "TT", "AG", "*", "CGA", "AT", "TC", "GA", "GT", "TG", "CAG", "ACG", "CCG", "AA",
                 "CTG", "GCG", "TCG", "CGC", "CGG", "CGT", "AAC"]


@pytest.mark.parametrize("n_entries", [14, 15, 20, 26, 27, 32])
def test_more_than_thirteen_contexts_are_counted(n_entries):
    """The reference takes any number of contexts (src/mod.c:204-326 parses what -c names).  A 32-bit reference word holds the bits of thirteen context classes:
    since round 6 a run with more builds its site indices in passes of thirteen (mm_freq_create) and the kernels that test a position's context in the reference
    word -- the tile pipeline, the fused kernel -- take classes 13 and up from the site word (k_stream_reads asks the site word for every class).  14 to 32
    DIFFERENT contexts, the `*` context among them at a class behind the first pass, mixed and regular reads, through all three paths against the oracle."""
    rng = np.random.default_rng(6100 + n_entries)
    ref = make_ref(rng, 200000)
    recs = [_random_read(rng, ref, 16 if rng.random() < 0.5 else 0) for _ in range(40)] + [_mixed_read(rng, ref) for _ in range(80)]
    c = _entries(MANY_CODES[:n_entries], MANY_CONTEXTS[:n_entries])
    assert len(set(MANY_CONTEXTS[:n_entries])) == n_entries
    both_ways(recs, ref, c)
    got, _ = hip_rows(recs, ref, c, force_fused=True)
    assert got == oracle_rows(recs, ref, c)
    # the contexts in another order: other classes land behind the first pass, the rows are the same
    perm = list(rng.permutation(n_entries))
    c2 = ",".join(c.split(",")[i] for i in perm)
    assert sorted(oracle_rows(recs, ref, c2)) == sorted(hip_rows(recs, ref, c2)[0]) == sorted(hip_rows(recs, ref, c2, stream_mode=1)[0])


def test_more_than_thirty_two_entries_are_refused():
    rng = np.random.default_rng(5)
    ref = make_ref(rng, 100000)
    with pytest.raises(Exception):
        hip_rows([_random_read(rng, ref, 0)], ref, _entries(MANY_CODES + ["2"], ["CG"]))   # 33 entries


def test_haplotype_tags_beyond_the_dense_planes_are_counted():
    """--haplotypes with tags the handle keeps no dense plane for (the default is 8 planes; a tag is a byte: `HP:C`, src/mod.c:181-202): 9 - 16, 62, 200
    and 255 count through the side lists -- the same rows as the oracle, `*` aggregates included, in every stream mode and with --insertions."""
    import minimod_amd
    from minimod_amd import synth
    ref = synth.reference(31, 2 << 20)
    b = synth.batch(ref, 0, 600, seed=123, n_reads_total=600, haplotypes=True, long_insertions=True, max_len=20000.0)
    tags = np.array(list(range(0, 17)) + [62, 200, 255], dtype=b["reads"]["hp"].dtype)
    b["reads"]["hp"] = tags[np.arange(len(b["reads"])) % len(tags)]
    for kw in (dict(haplotypes=True), dict(haplotypes=True, insertions=True)):
        orc = O.Oracle([("m", "CG")], [0.8], ["chrS"], **kw); orc.add_contig("chrS", ref); orc.process(b, threads=8)
        want = orc.rows()
        key = lambda r, io: sorted(zip(r["pos"].tolist(), r["strand"].tolist(), r["code"].tolist(), r[io].tolist(), r["hp"].tolist(), r["n_called"].tolist(), r["n_mod"].tolist()))
        assert len(want) > 5000 and {9, 16, 62, 200, 255} <= set(want["hp"].tolist())
        for mode in (0, 1):
            eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)], stream_mode=mode, side_capacity=8 << 20, **kw)
            eng.process(b)
            got = eng.finalize(); eng.close()
            assert key(got, "ins_offset") == key(want, "ins_off"), (kw, mode)
