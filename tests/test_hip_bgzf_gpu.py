"""Device-side BGZF inflate (include/minimod_bgzf.h, csrc/bgzf_kernels.hip.h) against zlib: the reference's BAM files block by
block, synthetic streams of every DEFLATE block type and shape, the CRC32 check, and what a damaged block's status says."""
import glob
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def inf():
    from minimod_amd import bgzf
    h = bgzf.Inflater(slots=2, max_blocks=4096, max_cbytes=64 << 20, max_obytes=192 << 20)
    yield h
    h.close()


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
    return c.compress(data) + c.flush()


def make_block(data, **kw):
    return (raw_deflate(data, **kw), len(data), zlib.crc32(data) & 0xFFFFFFFF)


def check(inf, blocks, want):
    got, status = inf.inflate(blocks)
    assert status.tolist() == [0] * len(blocks), status.tolist()
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, "block %d differs (first at %d of %d)" % (i, next(k for k in range(min(len(g), len(w))) if g[k] != w[k]) if len(g) == len(w) else -1, len(w))


def test_reference_bams_block_by_block(inf):
    from minimod_amd import bgzf
    files = sorted(glob.glob(os.path.join(HERE, "golden", "data", "*.bam")))
    assert files
    for f in files:
        blocks = bgzf.split_bgzf(open(f, "rb").read())
        want = [zlib.decompress(p, -15) for p, _, _ in blocks]
        for k in range(0, len(blocks), 4096):
            check(inf, blocks[k:k + 4096], want[k:k + 4096])


def test_every_block_type_and_shape(inf):
    rng = np.random.default_rng(11)
    datas = [b"", b"A", b"AB" * 32000 + b"x", bytes(rng.integers(0, 256, 65280, dtype=np.uint8)),
             bytes(rng.integers(0, 4, 65000, dtype=np.uint8)), b"\x00" * 65280, bytes(range(256)) * 255,
             bytes(rng.choice(np.frombuffer(b"ACGT,;0123456789", dtype=np.uint8), 60000)),
             b"".join(bytes([int(x)]) * int(n) for x, n in zip(rng.integers(0, 256, 3000), rng.integers(1, 40, 3000)))[:65280]]
    blocks, want = [], []
    for d in datas:
        for kw in (dict(level=6), dict(level=1), dict(level=9), dict(level=0), dict(level=6, strategy=zlib.Z_FIXED), dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY),
                   dict(level=6, strategy=zlib.Z_RLE), dict(level=9, wbits=-9)):
            blocks.append(make_block(d, **kw)); want.append(d)
    # several DEFLATE blocks in one BGZF block (sync flushes in between), a stored block in the middle
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = [bytes(rng.integers(0, 50, 9000, dtype=np.uint8)) for _ in range(5)]
    raw = b"".join(c.compress(p) + c.flush(zlib.Z_SYNC_FLUSH if i != 2 else zlib.Z_FULL_FLUSH) for i, p in enumerate(parts)) + c.flush()
    whole = b"".join(parts)
    blocks.append((raw, len(whole), zlib.crc32(whole))); want.append(whole)
    check(inf, blocks, want)


def test_long_codes_and_far_matches(inf):
    """Skewed symbol statistics give codes longer than the first-level tables (11 / 8 bits); matches at the window's far end."""
    rng = np.random.default_rng(5)
    p = 1.0 / (1.6 ** np.arange(256)); p /= p.sum()
    skew = bytes(rng.choice(256, 65000, p=p).astype(np.uint8))
    far = bytes(rng.integers(0, 256, 30000, dtype=np.uint8))
    far = far + bytes(rng.integers(0, 256, 2700, dtype=np.uint8)) + far[:32000]
    blocks = [make_block(skew, level=9), make_block(far[:65280], level=9), make_block(skew[:1000] * 60, level=9)]
    check(inf, blocks, [skew, far[:65280], (skew[:1000] * 60)])


def test_damaged_blocks_are_reported_not_decoded(inf):
    rng = np.random.default_rng(3)
    good = bytes(rng.integers(0, 16, 40000, dtype=np.uint8))
    payload, isize, crc = make_block(good)
    flipped = bytearray(payload); flipped[len(flipped) // 2] ^= 0x10
    blocks = [(payload, isize, crc), (payload, isize, crc ^ 1), (payload, isize - 1, crc), (payload[:len(payload) // 2], isize, crc),
              (bytes(flipped), isize, crc), (bytes(rng.integers(0, 256, 500, dtype=np.uint8)), 4000, 0), (payload, isize, crc)]
    got, status = inf.inflate(blocks)
    assert status[0] == 0 and status[6] == 0 and got[0] == good and got[6] == good
    assert status[1] == 9                      # CRC mismatch
    assert all(int(s) != 0 for s in status[1:6])


def test_damaged_streams_never_decode_to_something_else(inf):
    """3 000 damaged deflate streams in one launch (bit flips, truncations, spliced and random bytes): every block comes back
    with a status; status 0 means the bytes are exactly what zlib makes of the same stream."""
    rng = np.random.default_rng(17)
    bases = [bytes(rng.integers(0, 256, 20000, dtype=np.uint8)), bytes(rng.integers(0, 8, 30000, dtype=np.uint8)),
             (b"ACGT,12,3;" * 3000), bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), 40000)), b"\x00" * 5000]
    good = [raw_deflate(d, level=lv, strategy=st) for d in bases for lv, st in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY))]
    sizes = [len(d) for d in bases for _ in range(4)]
    blocks = []
    for k in range(3000):
        i = int(rng.integers(0, len(good)))
        p = bytearray(good[i])
        kind = int(rng.integers(0, 5))
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))): p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1: p = p[:int(rng.integers(1, len(p)))]
        elif kind == 2:
            j = int(rng.integers(0, len(good))); a = int(rng.integers(0, len(p))); p = p[:a] + bytearray(good[j][int(rng.integers(0, len(good[j]))):])
        elif kind == 3: p = bytearray(rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8).tobytes())
        else: p[:int(rng.integers(1, 12))] = rng.integers(0, 256, 1, dtype=np.uint8).tobytes() * 1   # header bits
        p = bytes(p[:65000])
        isize = sizes[i] if rng.random() < 0.8 else int(rng.integers(0, 65536))
        try:
            ref = zlib.decompressobj(-15).decompress(p, 70000)
        except zlib.error:
            ref = None
        crc = zlib.crc32(ref) if ref is not None and rng.random() < 0.9 else int(rng.integers(0, 1 << 32))
        blocks.append((p, isize, crc, ref))
    got, status = inf.inflate([(p, isize, crc) for p, isize, crc, _ in blocks])
    n_ok = 0
    for (p, isize, crc, ref), g, st in zip(blocks, got, status):
        if st == 0:
            n_ok += 1
            assert ref is not None and len(ref) >= isize and g == ref[:isize] and zlib.crc32(g) == crc
    assert 0 < n_ok < len(blocks)


def test_valid_streams_no_compressor_writes(inf):
    """tools/fuzz_inflate.py's generator, bounded: 512 valid DEFLATE streams made token by token (random complete Huffman codes up
    to 15 bits, code-length runs across the literal/distance boundary, one or no distance code, end-of-block-only blocks, every
    match length and distance).  zlib agrees with the generator on every one of them (checked here as well); the device returns
    those bytes -- a refusal (status != 0) would only send the block to the host decoder, but none is expected."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import fuzz_inflate
    streams = []
    for seed in range(900, 908):
        s, rejects = fuzz_inflate.make(seed, 64)
        assert rejects == 0
        streams += s
    assert all(zlib.decompress(raw, -15) == want for raw, want in streams)
    got, status = inf.inflate([(raw, len(want), zlib.crc32(want) & 0xFFFFFFFF) for raw, want in streams])
    assert status.tolist() == [0] * len(streams), [(i, int(s)) for i, s in enumerate(status) if s][:10]
    assert all(g == want for g, (_, want) in zip(got, streams))


def test_two_slots_in_flight_and_a_file_sized_launch(inf):
    """A C2-shape BAM's blocks, thousands per launch, two launches in flight; every decoded byte against zlib."""
    from minimod_amd import bgzf, synth
    import tempfile
    ref = synth.reference(3, 4 << 20)
    bs = [synth.batch(ref, i * 2048, 2048, seed=9, n_reads_total=4096, with_order=False) for i in range(2)]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "s.bam")
        synth.write_bam_parallel(path, [("chrS", len(ref))], bs, threads=4)
        blocks = bgzf.split_bgzf(open(path, "rb").read())
    want = [zlib.decompress(p, -15) for p, _, _ in blocks]
    half = len(blocks) // 2
    groups = [blocks[:half], blocks[half:]]
    sizes = [inf.fill(s, g) for s, g in enumerate(groups)]
    for s, (n, c, o) in enumerate(sizes):
        inf.submit(s, n, c, o)
    k = 0
    for s, (n, c, o) in enumerate(sizes):
        st = inf.wait(s, n)
        assert not st.any()
        out, br = inf.out_buffer(s), inf.blocks(s)
        for i in range(n):
            a = int(br[i]["o_off"])
            assert bytes(out[a:a + int(br[i]["isize"])]) == want[k], (s, i)
            k += 1
        t = inf.times(s)
        assert t and t["inflate_ms"] > 0
    assert k == len(blocks)


def test_cli_with_gpu_inflate_writes_the_same_bytes(tmp_path):
    """`minimod freq --gpu-inflate`: a reference golden (a file smaller than one group) and a synthetic BAM of several groups,
    byte for byte what the host pool alone gives (the default for files this small, and `--no-gpu-inflate`); the log says how
    many groups the device took."""
    import subprocess
    from minimod_amd import synth
    root = os.path.dirname(HERE)
    cli = os.path.join(root, "minimod_amd", "bin", "minimod")
    data = os.path.join(HERE, "golden", "data")
    ref = synth.reference(3, 8 << 20)
    bs = [synth.batch(ref, i * 2048, 2048, seed=4, n_reads_total=8192, with_order=False) for i in range(4)]
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    synth.write_bam_parallel(bam, [("chrS", len(ref))], bs, threads=4)
    synth.write_fasta(fa, "chrS", ref)
    cases = [(os.path.join(data, "example-ont.bam"), os.path.join(data, "hg38_chr22.fa") if os.path.exists(os.path.join(data, "hg38_chr22.fa")) else None),
             (bam, fa)]
    for b, f in cases:
        if f is None:
            continue
        outs = []
        for flags in ([], ["--gpu-inflate"], ["--no-gpu-inflate"]):
            out = str(tmp_path / ("o%d.bed" % len(outs)))
            r = subprocess.run([cli, "freq", "-b", "-c", "m[CG]", "-t", "4", "-o", out] + flags + [f, b], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            outs.append((open(out, "rb").read(), r.stderr.decode()))
        assert outs[0][0] == outs[1][0] == outs[2][0]
        assert "[gpu-inflate]" in outs[1][1]
        # without a flag the file's size decides (4 GiB per GPU): these are the host threads' alone
        assert "[gpu-inflate]" not in outs[0][1] and "[gpu-inflate]" not in outs[2][1]
        if b == bam:
            import re
            m = re.search(r"\[gpu-inflate\] (\d+) groups \((\d+) blocks\) inflated on the device, (\d+) blocks again on the host", outs[1][1])
            assert m and int(m.group(1)) >= 1 and int(m.group(3)) == 0, outs[1][1][-500:]


def test_view_and_devices_with_gpu_inflate(tmp_path):
    """`minimod view --gpu-inflate` and `minimod freq --devices 0,0 --gpu-inflate` (two workers on one GPU, an inflater each)
    write what they write without the flag."""
    import subprocess
    from minimod_amd import synth
    root = os.path.dirname(HERE)
    cli = os.path.join(root, "minimod_amd", "bin", "minimod")
    ref = synth.reference(5, 4 << 20)
    bs = [synth.batch(ref, i * 1024, 1024, seed=6, n_reads_total=3072, with_order=False) for i in range(3)]
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    synth.write_bam(bam, [("chrS", len(ref))], bs, index=True)
    synth.write_fasta(fa, "chrS", ref)
    assert os.path.exists(bam + ".bai")
    for cmd in (["view", "-c", "m[CG]", "-t", "4"], ["freq", "-b", "-c", "m[CG]", "-t", "4", "--devices", "0,0"]):
        outs = []
        for flags in ([], ["--gpu-inflate"]):
            out = str(tmp_path / ("o%d.txt" % len(outs)))
            r = subprocess.run([cli] + cmd + flags + ["-o", out, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            outs.append(open(out, "rb").read())
        assert outs[0] == outs[1] and len(outs[0]) > 1000, cmd


def test_cli_damaged_bam_fails_alike_with_and_without_the_device(tmp_path):
    """A BAM with one damaged BGZF block (a flipped CRC32 trailer; a flipped bit in the deflate payload; a block cut off by the
    file's end): the device refuses the block, the host decoder judges it -- `minimod freq --gpu-inflate` ends with the same exit
    code and the same last message as `--no-gpu-inflate`, and writes no rows for it."""
    import subprocess
    from minimod_amd import bgzf, synth
    root = os.path.dirname(HERE)
    cli = os.path.join(root, "minimod_amd", "bin", "minimod")
    ref = synth.reference(13, 4 << 20)
    bs = [synth.batch(ref, i * 1500, 1500, seed=8, n_reads_total=4500, with_order=False) for i in range(3)]
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    synth.write_bam(bam, [("chrS", len(ref))], bs)
    synth.write_fasta(fa, "chrS", ref)
    raw = open(bam, "rb").read()
    # block boundaries (offset, total size) from the BSIZE fields
    pos, blocks = 0, []
    while pos < len(raw):
        total = int.from_bytes(raw[pos + 16:pos + 18], "little") + 1
        blocks.append((pos, total))
        pos += total
    assert len(blocks) > 1200          # more than one group of 1024 blocks
    victims = [len(blocks) // 3, 1024 + 50]
    cases = {}
    for name, k in (("crc", victims[0]), ("payload", victims[1])):
        off, total = blocks[k]
        bad = bytearray(raw)
        if name == "crc":
            bad[off + total - 8] ^= 0x01
        else:
            bad[off + 18 + (total - 26) // 2] ^= 0x08
        cases[name] = bytes(bad)
    off, total = blocks[len(blocks) // 2]
    cases["cut"] = raw[:off + total // 2]
    for name, data in cases.items():
        p = str(tmp_path / (name + ".bam"))
        open(p, "wb").write(data)
        res = []
        for flag in ("--no-gpu-inflate", "--gpu-inflate"):
            out = str(tmp_path / (name + flag + ".bed"))
            r = subprocess.run([cli, "freq", "-b", "-c", "m[CG]", "-t", "4", flag, "-o", out, fa, p], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            msgs = [l for l in r.stderr.decode(errors="replace").splitlines() if "ERROR" in l or "error" in l]
            res.append((r.returncode, msgs[-1] if msgs else ""))
        assert res[0][0] != 0 and res[0] == res[1], (name, res)


def test_cli_reads_a_pipe_with_gpu_inflate_asked_for(tmp_path):
    """The BAM through a named pipe with `--gpu-inflate`: a pipe's bytes can be read once (no header read-ahead) and cannot be
    mapped (the host threads inflate alone) -- the rows are the file's."""
    import subprocess
    import threading
    from minimod_amd import synth
    root = os.path.dirname(HERE)
    cli = os.path.join(root, "minimod_amd", "bin", "minimod")
    ref = synth.reference(5, 2 << 20)
    bam, fa, pipe = str(tmp_path / "s.bam"), str(tmp_path / "s.fa"), str(tmp_path / "p.bam")
    synth.write_bam(bam, [("chrS", len(ref))], [synth.batch(ref, 0, 800, seed=6, n_reads_total=800, with_order=False)])
    synth.write_fasta(fa, "chrS", ref)
    a, b = str(tmp_path / "a.bed"), str(tmp_path / "b.bed")
    r = subprocess.run([cli, "freq", "-b", "-c", "m[CG]", "-t", "4", "-o", a, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    os.mkfifo(pipe)

    def feed():
        with open(pipe, "wb") as w:
            w.write(open(bam, "rb").read())
    t = threading.Thread(target=feed, daemon=True)
    t.start()
    r = subprocess.run([cli, "freq", "-b", "-c", "m[CG]", "-t", "4", "--gpu-inflate", "-o", b, fa, pipe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    t.join(timeout=10)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    assert open(a, "rb").read() == open(b, "rb").read() and os.path.getsize(a) > 1000
