"""CPU-only tests of the host side: the C ABI library loads and exports what include/minimod_hip.h declares, and the
C host code (options, FASTA, BGZF/BAM loader) agrees with the oracle's independent Python restatements."""
import os
import re

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pybam
from tests.cases import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from minimod_amd import engine
    L = engine.load_library()
    hdr = open(os.path.join(ROOT, "include", "minimod_hip.h")).read()
    declared = set(re.findall(r"\b(mm_[a-z_0-9]+)\s*\(", hdr))
    declared = {d for d in declared if not d.endswith("_t")}
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(L, name), "library does not export %s" % name
    assert declared == set(engine.EXPORTS)
    assert L.mm_abi_version() == engine.MM_ABI_VERSION
    assert engine.READ_DTYPE.itemsize == 64 == pybam.READ_DTYPE.itemsize
    assert [n for n in engine.READ_DTYPE.names] == [n for n in pybam.READ_DTYPE.names]


def test_allocator_symbols_exported():
    """mm_devmem_stats / mm_devmem_trim (include/minimod_bgzf.h; csrc/devmem.h): exported, and no unit of the device library goes to the
    driver's malloc / free by itself -- an address the driver takes back and hands out again is what broke the site index (round 6)."""
    import ctypes, glob
    from minimod_amd import build
    L = ctypes.CDLL(build.lib_path())
    hdr = open(os.path.join(ROOT, "include", "minimod_bgzf.h")).read()
    for name in ("mm_devmem_stats", "mm_devmem_trim"):
        assert name in hdr and hasattr(L, name), name
    out = (ctypes.c_int64 * 7)(*([-1] * 7))
    L.mm_devmem_stats(out)   # (bookkeeping only: no HIP call, so it answers without a GPU -- nothing held, nothing kept)
    assert list(out) == [0] * 7
    for f in glob.glob(os.path.join(ROOT, "minimod_amd", "csrc", "*.hip*")):
        src = open(f).read()
        assert not re.search(r"\bhip(Malloc|Free|HostMalloc|HostFree)\(", src), "%s allocates behind the library's allocator" % os.path.basename(f)


def test_bgzf_abi_symbols_exported():
    """Every function include/minimod_bgzf.h declares is exported by the device library (no calls: there is no GPU here)."""
    import ctypes, re
    from minimod_amd import bgzf, build
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "minimod_bgzf.h")).read()
    declared = set(re.findall(r"\b(mm_bgzf_[a-z_]+)\s*\(", hdr))
    assert declared == set(bgzf.EXPORTS)
    L = ctypes.CDLL(build.lib_path())
    for name in declared:
        assert hasattr(L, name), name
    assert bgzf.BLOCK_DTYPE.itemsize == 20


def test_ingest_abi_symbols_exported():
    """Every function include/minimod_ingest.h declares is exported by the device library, and the host library's device loader
    (csrc/host/devloader.c) is there (no calls: there is no GPU here)."""
    import ctypes, re
    from minimod_amd import build
    from minimod_amd.synth import host_lib
    hdr = open(os.path.join(ROOT, "include", "minimod_ingest.h")).read()
    declared = set(re.findall(r"\b(mm_ingest_[a-z_]+)\s*\(", hdr))
    declared = {d for d in declared if not d.endswith("_t")}
    assert len(declared) >= 15
    L = ctypes.CDLL(build.lib_path())
    for name in sorted(declared):
        assert hasattr(L, name), name
    H = host_lib()
    for name in ("mmh_devloader_open", "mmh_devloader_next", "mmh_devloader_release", "mmh_devloader_stream", "mmh_devloader_fetch", "mmh_devloader_stats", "mmh_devloader_close",
                 "mm_bam_peek_header2", "mm_bgzf_block_total", "mm_bgzf_inflate_host"):
        assert hasattr(H, name), name


def test_tie_abi_symbols_exported():
    """Every function include/minimod_tie.h declares is exported by the device library (no calls: there is no GPU here), and the host
    library has the serial checker beside it."""
    import ctypes, re
    from minimod_amd import build
    from minimod_amd.synth import host_lib
    hdr = open(os.path.join(ROOT, "include", "minimod_tie.h")).read()
    declared = set(re.findall(r"\b(mm_(?:tie|fmt)_[a-z_]+)\s*\(", hdr))
    declared = {d for d in declared if not d.endswith("_t")}
    assert len(declared) >= 10
    L = ctypes.CDLL(build.lib_path())
    for name in sorted(declared):
        assert hasattr(L, name), name
    H = host_lib()
    for name in ("mmh_tie_order_plain", "mmh_tie_add_batch", "mmh_tie_order_rows_mt", "mmh_tie_export", "mmh_tie_import"):
        assert hasattr(H, name), name


def test_summary_abi_symbols_exported():
    import ctypes, re
    from minimod_amd import build
    hdr = open(os.path.join(ROOT, "include", "minimod_summary.h")).read()
    declared = {d for d in set(re.findall(r"\b(mm_summary_[a-z_]+)\s*\(", hdr)) if not d.endswith("_t")}
    assert declared == {"mm_summary_create", "mm_summary_batch", "mm_summary_destroy"}
    L = ctypes.CDLL(build.lib_path())
    for name in sorted(declared):
        assert hasattr(L, name), name


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import minimod_amd
    with pytest.raises(minimod_amd.MinimodHipError):
        minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrT", 4, b"ACGT")])


OPT_CASES = ["m", "m[CG]", "m,h", "m[CG],h[CG]", "h", "*", "m[*]", "*[CG]", "21839[C],m[*]", "17802[*],a,m[C]", "e,b",
             "a[A]", "m[Ct]", "hm[CG]", "m[cg],h[*],a", "o,n,f,c,g,T,U,A,C,G,N"]


@pytest.mark.parametrize("c", OPT_CASES)
def test_mod_code_parser_matches_oracle(c):
    from minimod_amd import hostlib
    want = O.parse_mod_codes(c)
    got = hostlib.parse_mods(c, None)
    assert [(a, b) for a, b, _ in got] == want
    assert all(t == 0.8 for _, _, t in got)


def test_mod_code_parser_errors():
    from minimod_amd import hostlib
    for bad in ["m[CG", "m1", "m[C*]", "m,m", "m[X]", "m-"]:
        with pytest.raises(ValueError):
            O.parse_mod_codes(bad)
        with pytest.raises(ValueError):
            hostlib.parse_mods(bad)
    with pytest.raises(ValueError):
        hostlib.parse_mods("m,h", "0.8,0.7,0.6")
    with pytest.raises(ValueError):
        hostlib.parse_mods("m", "1.5")
    assert [t for _, _, t in hostlib.parse_mods("m,h", "0.8,0.5")] == [0.8, 0.5]
    assert [t for _, _, t in hostlib.parse_mods("m,h,a", "0.7")] == [0.7, 0.7, 0.7]


def test_threshold_classes_and_numbers():
    from minimod_amd import engine, hostlib
    for th in (0.0, 0.2, 0.5, 0.7, 0.8, 0.9, 0.999, 1.0):
        a, b = hostlib.klass_lut(th), engine.klass_lut(th)
        assert (a == b).all()
        for x in range(256):
            p = (x + 0.5) / 256.0
            want = 3 if p >= th else (1 if p <= 1 - th else 0)
            assert a[x] == want
    assert hostlib.parse_num("100M") == 100000000 and hostlib.parse_num("20K") == 20000 and hostlib.parse_num("1.5G") == 1500000000
    assert hostlib.parse_num("4096") == 4096


def test_fasta_loader(tmp_path):
    from minimod_amd import hostlib
    p = tmp_path / "r.fa"
    p.write_text(">chrA some description\nACGTacgt\nNNnn\n\n>chrB\tx\nUUuu\r\nGG\n>empty\n>chrC\nA")
    got = hostlib.load_ref(str(p))
    assert got == [("chrA", b"ACGTacgtNNnn"), ("chrB", b"UUuuGG"), ("empty", b""), ("chrC", b"A")]
    import gzip
    pz = tmp_path / "r.fa.gz"
    with gzip.open(str(pz), "wb") as f:
        f.write(p.read_bytes())
    assert hostlib.load_ref(str(pz)) == got


def test_fasta_parsers_agree(tmp_path):
    """The mapped parser (plain files, parallel passes) against the stream parser (what a .gz gets) on text that tries the rules:
    junk before the first record, '>' inside a line and after a carriage return, blanks inside sequence lines, a header the file
    ends in, empty records, a name of more than 1023 characters, records bigger than a piece and than a find chunk."""
    import numpy as np
    from minimod_amd import hostlib
    rng = np.random.default_rng(7)
    big = "\n".join("".join("ACGTNacgtn"[int(x)] for x in rng.integers(0, 10, 61)) for _ in range(200000))   # 12 MB
    cases = {
        "plain": ">a\nAC\nGT\n>b desc\nTT\n",
        "junk_first": "xx\nyy >no\n>a\nAC>GT\n\r>notarecord\nGG\n",
        "blanks": ">a\tz\nA C\tG\r\n \n\nT\n>b\r\nAA\n",
        "header_at_eof": ">a\nAC\n>b",
        "header_only_eof_nl": ">a\nAC\n>b\n",
        "empty_records": ">a\n>b\n>c\nA\n>d\n",
        "no_records": "ACGT\nACGT\n",
        "empty_file_ish": "\n\n",
        "long_name": ">" + "n" * 3000 + " d\nACGT\n",
        "no_final_newline": ">a\nACGT",
        "big": ">big one\n" + big + "\n>tail\nAC\n" + big[:100000],
    }
    for name, text in cases.items():
        p = tmp_path / (name + ".fa")
        p.write_bytes(text.encode())
        want = hostlib.load_ref(str(p), threads=0)
        for t in (1, 3, 8):
            assert hostlib.load_ref(str(p), threads=t) == want, (name, t)
    assert hostlib.load_ref(str(tmp_path / "plain.fa")) == [("a", b"ACGT"), ("b", b"TT")]
    assert [(n, len(s)) for n, s in hostlib.load_ref(str(tmp_path / "big.fa"), threads=4)] == [("big", 200000 * 61), ("tail", 2 + 100000 - big[:100000].count("\n"))]


def _same_batch(a, b):
    ra, rb = a["reads"], b["reads"]
    assert len(ra) == len(rb)
    for f in ("tid", "pos", "l_qseq", "n_cigar", "mm_len", "ml_len", "flag", "hp"):
        assert (ra[f] == rb[f]).all(), f
    for x, y in zip(ra, rb):
        n = int(x["n_cigar"])
        assert (a["cigar"][int(x["cigar_off"]):int(x["cigar_off"]) + n] == b["cigar"][int(y["cigar_off"]):int(y["cigar_off"]) + n]).all()
        n = (int(x["l_qseq"]) + 1) // 2
        assert (a["seq"][int(x["seq_off"]):int(x["seq_off"]) + n] == b["seq"][int(y["seq_off"]):int(y["seq_off"]) + n]).all()
        n = int(x["mm_len"])
        assert (a["mm"][int(x["mm_off"]):int(x["mm_off"]) + n + 1] == b["mm"][int(y["mm_off"]):int(y["mm_off"]) + n + 1]).all()
        n = int(x["ml_len"])
        assert (a["ml"][int(x["ml_off"]):int(x["ml_off"]) + n] == b["ml"][int(y["ml_off"]):int(y["ml_off"]) + n]).all()
        assert int(y["seq_off"]) % 16 == 0 and int(y["mm_off"]) % 16 == 0 and int(y["cigar_off"]) % 4 == 0


LOADER_CASES = [("example-ont.bam", dict(K=7)), ("example-hifi.bam", dict(K=1)), ("hap.bam", dict()),
                ("dna_5mCG_5hmCG_mm_with_secondary_chr22.bam", dict(allow_secondary=True)),
                ("dna_5mCG_5hmCG_mm_with_secondary_chr22.bam", dict(skip_supplementary=True, K=100)),
                ("dRNA.bam", dict(B=50000)), ("eb.bam", dict(K=4096))]


@pytest.mark.parametrize("bam,kw", LOADER_CASES, ids=["%s-%s" % (b, "-".join(k)) for b, k in LOADER_CASES])
def test_c_loader_matches_python_reader(bam, kw):
    """load_db in C (BGZF inflate pool + filters + flattening) == the oracle's pure-Python reader, batch by batch."""
    from minimod_amd import hostlib
    path = os.path.join(GOLDEN, "data", bam)
    got = list(hostlib.load_batches(path, **kw))
    want = [b for _, b, _ in pybam.load_batches(path, **kw)]
    assert len(got) == len(want)
    for a, b in zip(want, got):
        _same_batch(a, b)


def test_synthetic_generator_is_deterministic_and_valid():
    from minimod_amd import synth
    ref = synth.reference(7, 3 << 20)
    assert (synth.reference_slice(7, 1 << 20, 2 << 20) == ref[1 << 20:]).all()
    b1 = synth.batch(ref, 10, 50, seed=3, n_reads_total=500)
    b2 = synth.batch(ref, 0, 60, seed=3, n_reads_total=500)
    for k in ("l_qseq", "pos", "n_cigar", "mm_len", "flag"):
        assert (b1["reads"][k] == b2["reads"][k][10:]).all()
    rd = b1["reads"]
    assert (np.diff(rd["pos"]) >= 0).all()                       # coordinate sorted
    assert sorted(b1["order"].tolist()) == list(range(50))
    assert (rd["l_qseq"][b1["order"]][:-1] // 256 >= rd["l_qseq"][b1["order"]][1:] // 256).all()   # longest first
    # every read is well-formed for the oracle, and MM lists exactly the read's CpGs
    o = O.Oracle([("m", "CG"), ("h", "CG")], [0.8, 0.8], ["chrS"])
    o.add_contig("chrS", ref)
    o.process(b1)
    assert len(o.rows()) > 100
    for shape, dot in ((1, 0.5), (0, 1.0)):
        b = synth.batch(ref, 0, 20, seed=9, n_reads_total=200, shape=shape, dot_fraction=dot, haplotypes=True)
        o = O.Oracle([("m", "CG")], [0.8], ["chrS"], insertions=True, haplotypes=True)
        o.add_contig("chrS", ref)
        o.process(b)
        assert len(o.rows()) > 100


def test_plan_batch_splits_long_reads_and_orders_by_cost():
    from minimod_amd import engine, synth
    ref = synth.reference(3, 2 << 20)
    b = synth.batch(ref, 0, 300, seed=4, n_reads_total=300)
    items = engine.plan_batch(b["reads"])
    rd = b["reads"]
    ridx, part, nparts = items & 0xFFFFFF, (items >> 24) & 15, ((items >> 28) & 15) + 1
    # every read appears with all of its parts exactly once
    seen = {}
    for r, p_, n_ in zip(ridx.tolist(), part.tolist(), nparts.tolist()):
        seen.setdefault(r, []).append((p_, n_))
    assert sorted(seen) == list(range(len(rd)))
    for r, ps in seen.items():
        n_ = ps[0][1]
        assert sorted(p_ for p_, _ in ps) == list(range(n_)) and all(x == n_ for _, x in ps)
        assert n_ == min(16, max(1, -(-int(rd["l_qseq"][r]) // 24576)))
    cost = rd["l_qseq"][ridx] // nparts // 256
    assert (cost[:-1] >= cost[1:]).all()      # costliest first
    assert len(engine.plan_batch(rd[:0])) == 0


def test_synthetic_bam_roundtrip(tmp_path):
    """BGZF/BAM writer -> C loader and -> the oracle's Python reader: both give back the batch, filter fodder dropped."""
    from minimod_amd import hostlib, synth
    ref = synth.reference(9, 3 << 20)
    bs = [synth.batch(ref, i * 400, 400, seed=2, n_reads_total=800, haplotypes=True) for i in range(2)]
    p = str(tmp_path / "s.bam")
    synth.write_bam(p, [("chrS", len(ref))], bs)
    got = [g for g in hostlib.load_batches(p, K=400, B=10 ** 9) if len(g["reads"])]
    want = [b for _, b, _ in pybam.load_batches(p, K=400, B=10 ** 9) if len(b["reads"])]
    assert len(got) == len(want) == 2
    for a, b, c in zip(bs, got, want):
        _same_batch(a, b)
        _same_batch(a, c)
    n_records = sum(1 for _ in pybam.BamFile(p))
    assert n_records > 800      # the unmapped / secondary / tag-less copies are in the file


def test_multi_contig_bam_roundtrip_and_oracle_contig_order(tmp_path):
    """Reads on four contigs whose names sort differently by strcmp than by tid, batches that run across contig
    boundaries: BAM writer -> C loader == Python reader == the generated batches; the oracle's rows come out in strcmp
    order of the contig names (cmp_key_fast, reference src/mod.c:59-87) and do not depend on how the reads are batched."""
    from minimod_amd import hostlib, synth
    names, lens = ["chr1", "chr2", "chr10", "chrX", "chrM"], [1 << 19, 1 << 19, 1 << 18, 1 << 19, 16569]
    refs = [synth.reference(100 + i, L) for i, L in enumerate(lens[:4])] + [None]
    bs = synth.multi_contig(refs, [120, 90, 60, 100, 0], 128, seed=5, median_len=3000.0, max_len=20000.0)
    assert any(len(set(b["reads"]["tid"].tolist())) > 1 for b in bs)
    p = str(tmp_path / "g.bam")
    synth.write_bam(p, list(zip(names, lens)), bs, filter_fodder=True)
    got = [g for g in hostlib.load_batches(p, K=128, B=10 ** 9, threads=3) if len(g["reads"])]
    want = [b for _, b, _ in pybam.load_batches(p, K=128, B=10 ** 9) if len(b["reads"])]
    assert [len(g["reads"]) for g in got] == [len(b["reads"]) for b in bs] == [len(w["reads"]) for w in want]
    for a, b, c in zip(bs, got, want):
        assert (a["reads"]["tid"] == b["reads"]["tid"]).all() and (a["reads"]["pos"] == c["reads"]["pos"]).all()
        for k in ("l_qseq", "n_cigar", "mm_len", "ml_len", "flag"):
            assert (a["reads"][k] == b["reads"][k]).all() and (a["reads"][k] == c["reads"][k]).all()

    def rows(batches, threads):
        o = O.Oracle([("m", "CG"), ("h", "CG")], [0.8, 0.7], names)
        for n, r in zip(names, refs):
            if r is not None:
                o.add_contig(n, r)
        for b in batches:
            o.process(b, threads=threads)
        return o.rows()
    a, b = rows(bs, 3), rows(got, 1)
    assert len(a) > 5000 and (a == b).all()
    assert [names[t] for t in dict.fromkeys(a["tid"].tolist())] == ["chr1", "chr10", "chr2", "chrX"]


def _raw_batch(lengths):
    """Reads of the given lengths built straight into the flattened layout (random bases, one `N+m?` call each)."""
    rng = np.random.default_rng(5)
    n = len(lengths)
    reads = np.zeros(n, dtype=pybam.READ_DTYPE)
    cig, seq, mm, ml = [], [], [], []
    co = so = mo = lo = 0
    for i, L in enumerate(lengths):
        nib = np.array([1, 2, 4, 8], dtype=np.uint8)[rng.integers(0, 4, L + (L & 1))]
        if L & 1:
            nib[-1] = 0
        packed = (nib[0::2] << 4) | nib[1::2]
        text = b"N+m?,%d;" % (i % 7)
        r = reads[i]
        r["cigar_off"], r["seq_off"], r["mm_off"], r["ml_off"] = co, so, mo, lo
        r["tid"], r["pos"], r["l_qseq"], r["n_cigar"], r["mm_len"], r["ml_len"], r["flag"] = 0, 10 + i, L, 1, len(text), 1, 0
        c = np.zeros(4, dtype="<u4"); c[0] = (L << 4) | 0
        cig.append(c); co += 4
        s = np.zeros((len(packed) + 15) // 16 * 16, dtype=np.uint8); s[:len(packed)] = packed
        seq.append(s); so += len(s)
        m = np.zeros((len(text) + 1 + 15) // 16 * 16, dtype=np.uint8); m[:len(text)] = np.frombuffer(text, dtype=np.uint8)
        mm.append(m); mo += len(m)
        l = np.zeros(4, dtype=np.uint8); l[0] = 200 + i
        ml.append(l); lo += 4
    pad = lambda parts, dt: np.concatenate(parts + [np.zeros(64, dtype=dt)])
    return {"reads": reads, "cigar": pad(cig, "<u4"), "seq": pad(seq, np.uint8), "mm": pad(mm, np.uint8), "ml": pad(ml, np.uint8)}


def test_loader_record_bigger_than_head_room_and_many_chunks(tmp_path):
    """Records of 5, 5, 9 and 0.1 MB: the 9 MB one starts 6 MB before the end of the first 16 MiB chunk of decoded stream,
    more than the reader's 4 MiB head room (the spill path of bamio.c); then a 60 MB stream of ordinary reads (several
    chunks, records straddling them through the head room).  1 and 5 worker threads; same batches as the Python reader."""
    from minimod_amd import hostlib, synth
    b = _raw_batch([50, 3_300_000, 3_300_000, 6_000_000, 70_000, 33])
    p = str(tmp_path / "giant.bam")
    synth.write_bam(p, [("chrT", 8_000_000)], [b], filter_fodder=False)
    want = [x for _, x, _ in pybam.load_batches(p, K=4, B=10 ** 9)]
    for th in (1, 5):
        got = list(hostlib.load_batches(p, K=4, B=10 ** 9, threads=th))
        assert [len(g["reads"]) for g in got] == [4, 2] == [len(w["reads"]) for w in want]
        for a, c in zip(want, got):
            _same_batch(a, c)
    ref = synth.reference(9, 4 << 20)
    bs = [synth.batch(ref, i * 700, 700, seed=4, n_reads_total=2100) for i in range(3)]
    p2 = str(tmp_path / "multi.bam")
    synth.write_bam(p2, [("chrS", len(ref))], bs)
    want = [x for _, x, _ in pybam.load_batches(p2, K=333, B=10 ** 9) if len(x["reads"])]
    for th in (1, 5):
        got = [g for g in hostlib.load_batches(p2, K=333, B=10 ** 9, threads=th) if len(g["reads"])]
        assert len(got) == len(want) and sum(len(g["reads"]) for g in got) == 2100
        for a, c in zip(want, got):
            _same_batch(a, c)


def test_loader_reports_truncated_file(tmp_path):
    from minimod_amd import hostlib, synth
    ref = synth.reference(9, 1 << 20)
    p = str(tmp_path / "t.bam")
    synth.write_bam(p, [("chrS", len(ref))], [synth.batch(ref, 0, 200, seed=4, n_reads_total=200)])
    raw = open(p, "rb").read()
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(raw[:len(raw) * 2 // 3])
    with pytest.raises(IOError):
        list(hostlib.load_batches(cut, K=4096, B=10 ** 9, threads=3))


def _bgzf_blocks(raw):
    """(offset, total size) of every BGZF block of a file image."""
    out, pos = [], 0
    while pos < len(raw):
        xlen = int.from_bytes(raw[pos + 10:pos + 12], "little")
        assert raw[pos + 12:pos + 14] == b"BC"
        total = int.from_bytes(raw[pos + 16:pos + 18], "little") + 1
        out.append((pos, total, xlen))
        pos += total
    return out


def test_loader_checks_block_crc32(tmp_path):
    """A BGZF block whose CRC32 trailer does not match its payload fails the file, as with htslib: here a flipped CRC and a
    payload whose stored-block bytes were altered (same ISIZE, inflates fine).  The CRC itself equals zlib's on every length."""
    import ctypes
    import zlib
    from minimod_amd import hostlib, synth
    L = synth.host_lib()
    L.mm_crc32.restype = ctypes.c_uint32
    L.mm_crc32.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    rng = np.random.default_rng(3)
    blob = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    for n in list(range(0, 200)) + [255, 256, 4095, 4096, 65280, 65536, 70000]:
        assert L.mm_crc32(blob[:n], n) == zlib.crc32(blob[:n])
    ref = synth.reference(9, 1 << 20)
    p = str(tmp_path / "t.bam")
    synth.write_bam(p, [("chrS", len(ref))], [synth.batch(ref, 0, 300, seed=4, n_reads_total=300)])
    raw = bytearray(open(p, "rb").read())
    assert sum(len(b["reads"]) for b in hostlib.load_batches(p, K=4096, B=10 ** 9, threads=3)) == 300
    blocks = _bgzf_blocks(bytes(raw))
    assert len(blocks) > 4
    off, total, _ = blocks[2]
    bad = bytearray(raw)
    bad[off + total - 8] ^= 0x01                      # CRC32 trailer
    q = str(tmp_path / "crc.bam")
    open(q, "wb").write(bad)
    for env in ({}, {"MM_BAM_NO_MMAP": "1"}):       # mapped file and fread path
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            with pytest.raises(IOError):
                list(hostlib.load_batches(q, K=4096, B=10 ** 9, threads=3))
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    # a block re-deflated from altered bytes, ISIZE and CRC left as they were
    off, total, xlen = blocks[1]
    payload = bytearray(zlib.decompress(bytes(raw[off + 12 + xlen:off + total - 8]), -15))
    payload[len(payload) // 2] ^= 0x40
    comp = zlib.compressobj(1, zlib.DEFLATED, -15)
    cdata = comp.compress(bytes(payload)) + comp.flush()
    nb = bytearray(raw[off:off + 12 + xlen]) + cdata + raw[off + total - 8:off + total]
    nb[16:18] = (len(nb) - 1).to_bytes(2, "little")
    q2 = str(tmp_path / "payload.bam")
    open(q2, "wb").write(bytes(raw[:off]) + bytes(nb) + bytes(raw[off + total:]))
    with pytest.raises(IOError):
        list(hostlib.load_batches(q2, K=4096, B=10 ** 9, threads=2))


def test_aux_walk_rejects_payloads_that_leave_the_record():
    """mm_aux_get (bam_aux_get): a B array longer than the record, a Z string without its NUL or an unknown type end the
    walk with "no such tag"; well-formed tags behind well-formed ones are found."""
    import ctypes
    from minimod_amd import synth
    L = synth.host_lib()
    L.mm_aux_get.restype = ctypes.c_void_p
    L.mm_aux_get.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_char_p]

    def get(aux, tag):
        buf = ctypes.create_string_buffer(aux, len(aux))
        r = L.mm_aux_get(buf, len(aux), tag)
        return None if not r else r - ctypes.addressof(buf)
    good = b"NMC\x05" + b"MMZC+m?,1;\x00" + b"MLBC\x01\x00\x00\x00\xc8" + b"HPC\x02"
    assert get(good, b"MM") == 6 and get(good, b"ML") == 17 and get(good, b"HP") == len(good) - 2 and get(good, b"XX") is None
    assert get(b"MMZC+m?,1;", b"MM") is None                                  # no NUL inside the record
    assert get(b"MLBC\xff\xff\xff\x7f\xc8", b"ML") is None                 # 2 G entries in a 9-byte record
    assert get(b"MLBC\x02\x00\x00\x00\xc8", b"ML") is None                 # two entries promised, one present
    assert get(b"HPi\x01\x00", b"HP") is None                                 # a 4-byte integer cut off after two
    assert get(b"XYq\x00" + good, b"MM") is None                               # unknown type: the walk cannot continue
    assert get(b"MLBq\x01\x00\x00\x00\xc8", b"ML") is None                 # unknown array subtype


def test_two_loaders_in_one_process_do_not_share_state(tmp_path):
    from minimod_amd import hostlib, synth
    ref = synth.reference(9, 1 << 20)
    pa, pb = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    ba = [synth.batch(ref, i * 150, 150, seed=4, n_reads_total=300) for i in range(2)]
    bb = [synth.batch(ref, i * 150, 150, seed=8, n_reads_total=300, shape=1) for i in range(2)]
    synth.write_bam(pa, [("chrS", len(ref))], ba, filter_fodder=False)
    synth.write_bam(pb, [("chrS", len(ref))], bb, filter_fodder=False)
    ga, gb = hostlib.load_batches(pa, K=150, B=10 ** 9), hostlib.load_batches(pb, K=150, B=10 ** 9)
    for a, b in zip(ba, bb):              # interleaved: each generator owns an open loader
        _same_batch(a, next(ga))
        _same_batch(b, next(gb))
    ga.close(); gb.close()


SUMMARY_CASES = [
    ("test18.tsv", "dRNA.bam", []),
    ("dna_5mCG_5hmCG_mm_with_secondary_chr22_summary.tsv", "dna_5mCG_5hmCG_mm_with_secondary_chr22.bam", []),
    ("dna_5mCG_5hmCG_mm_with_secondary_chr22_summary_sec.tsv", "dna_5mCG_5hmCG_mm_with_secondary_chr22.bam", ["--allow-secondary"]),
    ("dna_5mCG_5hmCG_mm_with_secondary_chr22_summary_nosup.tsv", "dna_5mCG_5hmCG_mm_with_secondary_chr22.bam", ["--skip-supplementary"]),
    ("dna_5mCG_5hmCG_mm_with_secondary_chr22_summary_sec_nosup.tsv", "dna_5mCG_5hmCG_mm_with_secondary_chr22.bam",
     ["--allow-secondary", "--skip-supplementary"]),
]


@pytest.mark.parametrize("exp,bam,extra", SUMMARY_CASES, ids=[c[0] for c in SUMMARY_CASES])
def test_cli_summary_matches_reference_golden(exp, bam, extra):
    """`minimod summary` (host only, no GPU needed): byte-identical to the reference's goldens, which the reference's tests
    compare with a plain diff (test/test.sh:252-256,494-503) -- including the order of a read's keys, which is the slot
    order of the reference's hash table."""
    import subprocess
    import minimod_amd
    minimod_amd.build_all()
    binp = os.path.join(os.path.dirname(GOLDEN), "..", "minimod_amd", "bin", "minimod")
    for more in ([], ["-K", "7", "-t", "3"]):
        r = subprocess.run([binp, "summary"] + extra + more + [os.path.join(GOLDEN, "data", bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert r.stdout.decode() == open(os.path.join(GOLDEN, "expected", exp)).read()


@pytest.mark.parametrize("mode", [dict(), dict(bedmethyl=True), dict(insertions=True, haplotypes=True)],
                         ids=["tsv", "bedmethyl", "ins_hap"])
def test_c_row_formatter_serial_and_parallel_match_the_oracle_formatter(tmp_path, mode):
    """print_freq_output in C, formatted by a worker pool in pieces and written in order by the writer thread, is
    byte-identical to the serial run and to the oracle's formatter (which is pinned on the reference's goldens)."""
    from minimod_amd import engine, hostlib
    rng = np.random.default_rng(5)
    n = 200_000                                            # ~50 pieces: several rounds at 8 threads
    rows = np.zeros(n, dtype=engine.ROW_DTYPE)
    rows["tid"] = np.sort(rng.integers(0, 3, size=n))
    rows["pos"] = rng.integers(0, 250_000_000, size=n)
    rows["strand"] = rng.integers(0, 2, size=n)
    rows["code"] = rng.integers(0, 3, size=n)
    rows["n_called"] = np.where(rng.random(n) < 0.02, rng.integers(256, 100000, size=n), rng.integers(1, 256, size=n))
    rows["n_mod"] = (rows["n_called"] * rng.random(n)).astype(np.uint32)
    if mode.get("insertions"):
        rows["ins_offset"] = np.where(rng.random(n) < 0.1, rng.integers(1, 500, size=n), 0)
    rows["hp"] = rng.integers(-1, 3, size=n) if mode.get("haplotypes") else -1
    names, codes = ["chr1", "chr22_KI270731v1_random", "chrM"], ["m", "h", "21839"]
    outs = []
    for threads in (1, 8):
        path = str(tmp_path / ("rows_t%d.txt" % threads))
        hostlib.format_freq_rows(rows, names, codes, path, threads=threads, **mode)
        outs.append(open(path).read())
    assert outs[0] == outs[1]
    orows = np.zeros(n, dtype=O.ROW_DTYPE)
    for a, b in (("tid", "tid"), ("pos", "pos"), ("strand", "strand"), ("code", "code"), ("ins_off", "ins_offset"), ("hp", "hp"),
                 ("n_called", "n_called"), ("n_mod", "n_mod")):
        orows[a] = rows[b]
    assert outs[0] == O.format_rows(orows, names, codes, **mode)


def test_own_inflate_matches_zlib_on_every_block_type_and_rejects_damage():
    """inflate_fast.c (the BGZF reader's DEFLATE decoder) against zlib: stored, fixed and dynamic blocks, long and short
    matches, literal-heavy and run-heavy data, empty input; truncated streams and wrong sizes are rejected."""
    import zlib
    from minimod_amd import hostlib
    rng = np.random.default_rng(12)
    makers = [
        lambda n: rng.integers(0, 256, size=n, dtype=np.uint8).tobytes(),                          # incompressible -> stored
        lambda n: rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tobytes(),            # two bits of entropy
        lambda n: np.repeat(rng.integers(0, 256, size=n // 50 + 1, dtype=np.uint8), 50)[:n].tobytes(),   # runs (distance 1)
        lambda n: (rng.integers(33, 75, size=n, dtype=np.uint8)).tobytes(),                        # quality-like literals
        lambda n: (bytes(rng.integers(0, 256, size=400, dtype=np.uint8)) * (n // 400 + 1))[:n],    # far, long matches
    ]
    n_cases = 0
    for mk in makers:
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                for n in (0, 1, 2, 300, 65280, int(rng.integers(1000, 65536))):
                    raw = mk(n)
                    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
                    comp = co.compress(raw) + co.flush()
                    assert hostlib.inflate_raw(comp, len(raw)) == raw
                    n_cases += 1
                    if n > 300:
                        assert hostlib.inflate_raw(comp[:len(comp) // 2], len(raw)) is None      # truncated
                        assert hostlib.inflate_raw(comp, len(raw) - 1) is None                    # more data than room
                        assert hostlib.inflate_raw(comp, len(raw) + 1) is None                    # less data than promised
    assert n_cases == len(makers) * 4 * 4 * 6
    # the bundled BAMs, block by block
    for name in ("example-ont.bam", "hap.bam", "dRNA.bam"):
        d = open(os.path.join(GOLDEN, "data", name), "rb").read()
        p = 0
        while p + 18 < len(d):
            xlen = d[p + 10] | (d[p + 11] << 8)
            bsize = (d[p + 16] | (d[p + 17] << 8)) + 1
            comp = d[p + 12 + xlen:p + bsize - 8]
            isize = int.from_bytes(d[p + bsize - 4:p + bsize], "little")
            assert hostlib.inflate_raw(comp, isize) == zlib.decompress(comp, -15)
            p += bsize


@pytest.mark.parametrize("pieces", [False, True], ids=["whole", "pieces"])
def test_bai_linear_index_and_positioned_reader(tmp_path, pieces):
    """The writer's .bai (SAM specification 5.2) and the reader's use of it: for a set of (contig, position) starts, opening
    the BAM at mm_bai_start()'s virtual offset and skipping the records in front of the start gives exactly the records an
    unpositioned read of the file gives from there on.  pieces: the file written in rounds of pieces (bench.py --e2e-gbases), its
    index the pieces' indices shifted and joined (synth.merge_bai)."""
    import ctypes
    from minimod_amd import hostlib, synth
    names, lens = ["chr1", "chr2", "chr10"], [1 << 20, 3 << 19, 1 << 19]
    refs = [synth.reference(100 + i, L) for i, L in enumerate(lens)]
    bs = synth.multi_contig(refs, [300, 400, 150], 256, seed=5, median_len=4000.0, max_len=30000.0)
    p = str(tmp_path / "i.bam")
    if not pieces:
        synth.write_bam(p, list(zip(names, lens)), bs, index=True)
    else:
        parts, base, first = [], 0, 0
        with open(p, "wb") as out:
            for r0 in range(0, len(bs), 2):
                rp = str(tmp_path / "round.bam")
                pr = synth.write_bam_rounds(rp, list(zip(names, lens)), bs[r0:r0 + 2], first_round=r0 == 0, last_round=r0 + 2 >= len(bs), first_read=first, threads=2, index=True)
                parts += [(x, base + at) for x, at in pr]
                base += os.path.getsize(rp)
                first += sum(len(b["reads"]) for b in bs[r0:r0 + 2])
                out.write(open(rp, "rb").read())
        assert len(parts) == len(bs) and len(bs) > 2
        synth.merge_bai(p + ".bai", parts)
    L = hostlib._lib()
    L.mm_bai_load.restype = ctypes.c_void_p
    L.mm_bai_load.argtypes = [ctypes.c_char_p]
    L.mm_bai_start.restype = ctypes.c_uint64
    L.mm_bai_start.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64]
    L.mm_bai_free.argtypes = [ctypes.c_void_p]
    L.mmh_loader_open_share.restype = ctypes.c_void_p
    L.mmh_loader_open_share.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                        ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    bai = L.mm_bai_load((p + ".bai").encode())
    assert bai
    allr = np.concatenate([b["reads"] for b in bs])
    key = allr["tid"].astype(np.int64) << 32 | allr["pos"].astype(np.int64)
    from minimod_amd.engine import READ_DTYPE, mm_batch_t
    for (lo_t, lo_p), (hi_t, hi_p) in [((0, 0), (0, 400000)), ((0, 400000), (1, 65536)), ((1, 65536), (1, 1 << 20)), ((1, 1 << 20), (2, 1 << 19)),
                                       ((2, 300000), (3, 0))]:
        v = L.mm_bai_start(bai, lo_t, lo_p)
        ld = L.mmh_loader_open_share(p.encode(), 2, 4096, 10 ** 9, 0, 0, v, lo_t, lo_p, hi_t, hi_p, 0, 0)
        assert ld
        got, more = [], ctypes.c_int(1)
        while more.value:
            b = mm_batch_t()
            n = L.mmh_loader_next(ld, 0, ctypes.byref(b), ctypes.byref(more))
            assert n >= 0
            if n:
                rd = np.frombuffer((ctypes.c_char * (n * 64)).from_address(b.reads), dtype=READ_DTYPE).copy()
                got.append(rd)
        L.mmh_loader_close(ld)
        got = np.concatenate(got) if got else np.zeros(0, dtype=READ_DTYPE)
        want = allr[(key >= (lo_t << 32 | lo_p)) & (key < (hi_t << 32 | hi_p))]
        assert len(want) > 0 and len(got) == len(want)
        for k in ("tid", "pos", "l_qseq", "n_cigar", "mm_len", "flag"):
            assert (got[k] == want[k]).all()
    assert L.mm_bai_start(bai, 3, 0) == 2 ** 64 - 1
    L.mm_bai_free(bai)


def test_peek_header_reads_what_the_reader_reads(tmp_path):
    """mm_bam_peek_header (bamio.h): the header alone with plain file reads -- every golden BAM, a synthetic one whose header
    spans several BGZF blocks (3 000 contigs with long names), and files that are not BAM or end inside the header."""
    import ctypes, glob, gzip, struct
    from minimod_amd.build import lib_path
    from oracle import pybam
    L = ctypes.CDLL(lib_path("libminimod_host.so"))

    class Hdr(ctypes.Structure):
        _fields_ = [("n_targets", ctypes.c_int32), ("target_name", ctypes.POINTER(ctypes.c_char_p)), ("target_len", ctypes.POINTER(ctypes.c_uint32))]
    L.mm_bam_peek_header.argtypes = [ctypes.c_char_p, ctypes.POINTER(Hdr)]
    L.mm_bam_hdr_free.argtypes = [ctypes.POINTER(Hdr)]

    def peek(path):
        h = Hdr()
        if L.mm_bam_peek_header(str(path).encode(), ctypes.byref(h)) != 0:
            return None
        out = ([h.target_name[i].decode() for i in range(h.n_targets)], [int(h.target_len[i]) for i in range(h.n_targets)])
        L.mm_bam_hdr_free(ctypes.byref(h))
        return out

    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "data", "*.bam")))
    assert files
    for f in files:
        want = pybam.BamFile(f)
        assert peek(f) == (want.target_name, want.target_len), f
    # a header of several blocks: Python's gzip writes ONE member, so the BGZF framing is made here (blocks of 60 000 bytes)
    names = ["contig_%05d_%s" % (i, "x" * (i % 90)) for i in range(3000)]
    text = b"@HD\tVN:1.6\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (n.encode(), 1000 + i) for i, n in enumerate(names))
    raw = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(names))
    raw += b"".join(struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", 1000 + i) for i, n in enumerate(names))
    import zlib

    def bgzf(data):
        out = b""
        for k in range(0, len(data), 60000):
            d = data[k:k + 60000]
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            p = c.compress(d) + c.flush()
            out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(p) + 25) + p + struct.pack("<II", zlib.crc32(d), len(d))
        return out
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    big = tmp_path / "big_header.bam"
    big.write_bytes(bgzf(raw) + eof)
    assert len(raw) > 3 * 60000
    assert peek(big) == (names, [1000 + i for i in range(len(names))])
    cut = tmp_path / "cut.bam"
    cut.write_bytes(bgzf(raw[:100000]) + eof)
    assert peek(cut) is None
    notbam = tmp_path / "x.bam"
    notbam.write_bytes(bgzf(b"SAM\1" + raw[4:]))
    assert peek(notbam) is None
    notbam.write_bytes(b"hello, world" * 10)
    assert peek(notbam) is None
    assert peek(tmp_path / "missing.bam") is None


def test_loader_hands_out_the_reads_in_front_of_a_damaged_block(tmp_path):
    """A damaged block fails the file when the reader gets THERE (htslib's behaviour): the blocks in front of it in the same group
    are good data.  Here: the reads decoded before the error are a prefix of the intact file's, the same whether the damage is
    a CRC, or the file's end inside a block -- and there are some."""
    from minimod_amd import hostlib, synth
    ref = synth.reference(9, 1 << 20)
    p = str(tmp_path / "t.bam")
    synth.write_bam(p, [("chrS", len(ref))], [synth.batch(ref, 0, 400, seed=5, n_reads_total=400)])
    raw = open(p, "rb").read()
    blocks = _bgzf_blocks(raw)
    assert 8 < len(blocks) < 250            # one group
    want = [int(x) for b in hostlib.load_batches(p, K=16, B=10 ** 9, threads=3) for x in b["reads"]["pos"]]
    assert len(want) == 400
    k = len(blocks) * 2 // 3
    off, total, _ = blocks[k]
    bad = bytearray(raw); bad[off + total - 8] ^= 0x01
    magic = bytearray(raw); magic[off + 1] ^= 0x10
    files = {"crc": bytes(bad), "cut": raw[:off + total // 2], "magic": bytes(magic)}
    for name, data in files.items():
        q = str(tmp_path / (name + ".bam"))
        open(q, "wb").write(data)
        for env in ({}, {"MM_BAM_NO_MMAP": "1"}):       # mapped file and fread path
            old = {k_: os.environ.get(k_) for k_ in env}
            os.environ.update(env)
            try:
                got = []
                with pytest.raises(IOError):
                    for b in hostlib.load_batches(q, K=16, B=10 ** 9, threads=3):
                        got += [int(x) for x in b["reads"]["pos"]]
                assert 100 < len(got) < 400 and got == want[:len(got)], (name, env, len(got))
            finally:
                for k_, v in old.items():
                    os.environ.pop(k_, None) if v is None else os.environ.__setitem__(k_, v)


def test_stream_kernel_fits_its_waves():
    """The built code objects' metadata (tools/resources.py): the lean and kIns instantiations of k_stream_reads keep the 72 VGPRs
    and the LDS that seven wavefronts a SIMD allow, the `.`-capable ones the 80 of six (DESIGN §4's kernel table quotes these)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("mm_resources", os.path.join(root, "tools", "resources.py"))
    res = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(res)
    seen = 0
    for k in (0, 1, 2):
        obj = os.path.join(root, "minimod_amd", "lib", "obj", "freq_api_k%d.o" % k)
        if not os.path.exists(obj):
            pytest.skip("no freq objects beside the library (a library built elsewhere)")
        for name, f in res.kernels(obj):
            if "k_stream_reads" not in name:
                continue
            seen += 1
            lean = f["group_segment_fixed_size"] <= 22592
            assert f["vgpr_count"] <= (72 if lean else 80), (name, f)
            assert f["group_segment_fixed_size"] * (7 if lean else 6) <= 160 * 1024, (name, f)
    assert seen == 36


def test_a_process_leaves_its_teardown_behind_without_holding_its_pipes():
    """csrc/host/exitpath.c: the helper that shares a dying process's address space must not hold the process's stdout / stderr (a caller that reads them to
    their end would wait for the helper), must see the process go and must go itself; a process that never touched the GPU makes no helper at all."""
    import subprocess
    import sys
    import time
    from minimod_amd import build as B
    child = ("import ctypes, os, sys\n"
             "L = ctypes.CDLL(%r)\n"
             "big = bytearray(200 << 20)\n"
             "for i in range(0, len(big), 4096): big[i] = 1\n"
             "ctypes.c_int.in_dll(L, 'mmh_gpu_in_use').value = int(sys.argv[1])\n"
             "if len(sys.argv) > 3 and sys.argv[3] == 'high':\n"   # the pipe's descriptors above 4096 (the helper kept its own write end there)
             "    import resource\n"
             "    resource.setrlimit(resource.RLIMIT_NOFILE, (8192, resource.getrlimit(resource.RLIMIT_NOFILE)[1]))\n"
             "    null = os.open('/dev/null', os.O_RDONLY)\n"
             "    for i in range(4300): os.dup(null)\n"
             "sys.stdout.write('the last word'); sys.stdout.flush(); sys.stderr.write('and its echo'); sys.stderr.flush()\n"
             "L.mmh_leave_teardown_behind()\n"
             "os._exit(7)\n") % os.path.join(B.LIBDIR, "libminimod_host.so")

    def helpers_of(pid_text):
        n = 0
        for p in os.listdir("/proc"):
            if p.isdigit():
                try:
                    with open("/proc/%s/cmdline" % p, "rb") as f:
                        cmd = f.read()
                    with open("/proc/%s/stat" % p) as f:
                        state = f.read().rsplit(") ", 1)[1][0]
                except OSError:
                    continue
                if pid_text.encode() in cmd and state != "Z":
                    n += 1
        return n
    on = dict(os.environ, MM_ASYNC_EXIT="1")   # (round 6: the helper is opt-in)
    on.pop("MM_SYNC_EXIT", None)
    for in_use, where in ((1, "low"), (1, "high"), (0, "low")):
        tag = "exitpath-test-%d-%d-%s" % (os.getpid(), in_use, where)
        t0 = time.time()
        r = subprocess.run([sys.executable, "-c", child, str(in_use), tag, where], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60, env=on)
        assert r.returncode == 7 and r.stdout == b"the last word" and r.stderr == b"and its echo"
        assert time.time() - t0 < 30
        deadline = time.time() + 20
        while helpers_of(tag) and time.time() < deadline:   # (the helper shares the child's command line; it leaves when it has seen the pipe close)
            time.sleep(0.05)
        assert helpers_of(tag) == 0
    # the default, and MM_SYNC_EXIT=1 on top of MM_ASYNC_EXIT=1: no helper
    off = {k: v for k, v in os.environ.items() if k not in ("MM_ASYNC_EXIT", "MM_SYNC_EXIT")}
    for env in (off, dict(on, MM_SYNC_EXIT="1")):
        r = subprocess.run([sys.executable, "-c", child, "1", "exitpath-test-sync-%d" % os.getpid()], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60, env=env)
        assert r.returncode == 7 and r.stdout == b"the last word"
        assert helpers_of("exitpath-test-sync-%d" % os.getpid()) == 0
