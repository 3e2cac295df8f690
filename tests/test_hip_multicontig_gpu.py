"""Reads on SEVERAL contigs (BASELINE.json configs[3], the whole-genome shape, scaled down): contig order of the output is
strcmp on the names (cmp_key_fast, reference src/mod.c:59-87), not BAM-header order -- `chr1 < chr10 < chr2 < chrX` -- and
a -K batch of a coordinate-sorted BAM runs across contig boundaries.  Library path (site-major K2, the per-run merge path,
haplotype planes, the side list), the CLI (bedmethyl byte-identical to the oracle's), view, and a contig-sharded run of
two handles.  The reference's production invocation of this shape is test/test_ext.sh:60-70."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "minimod_amd", "bin", "minimod")

# BAM-header (tid) order differs from strcmp order; chrM is in the header but neither in the FASTA nor hit by a read
NAMES = ["chr1", "chr2", "chr10", "chrX", "chrM"]
LENS = [1 << 20, 3 << 19, 1 << 19, 1 << 20, 16569]
NREADS = [300, 400, 150, 250, 0]


@pytest.fixture(scope="module")
def genome():
    from minimod_amd import synth
    refs = [synth.reference(100 + i, L) for i, L in enumerate(LENS[:4])] + [None]
    return refs


def _batches(genome, batch_reads=256, **kw):
    from minimod_amd import synth
    kw.setdefault("median_len", 4000.0)
    kw.setdefault("max_len", 30000.0)
    return synth.multi_contig(genome, NREADS, batch_reads, seed=5, **kw)


def _oracle(genome, batches, mods, th, **kw):
    orc = O.Oracle(mods, th, NAMES, **kw)
    for n, r in zip(NAMES, genome):
        if r is not None:
            orc.add_contig(n, r)
    for b in batches:
        orc.process(b, threads=8)
    return orc


def _engine(genome, mods, th, **kw):
    import minimod_amd
    ctg = [(n, l, r) for n, l, r in zip(NAMES, LENS, genome)]
    return minimod_amd.FreqEngine([(c, x, t) for (c, x), t in zip(mods, th)], ctg, **kw)


def _key_rows(rows, io):
    return list(zip(rows["tid"].tolist(), rows["pos"].tolist(), rows["strand"].tolist(), rows["code"].tolist(), rows[io].tolist(),
                    rows["hp"].tolist(), rows["n_called"].tolist(), rows["n_mod"].tolist()))


CASES = [
    ("m_site_major", [("m", "CG")], [0.8], dict(), dict()),
    ("mh_site_major", [("m", "CG"), ("h", "CG")], [0.8, 0.7], dict(), dict()),
    ("m_by_runs", [("m", "CG")], [0.8], dict(), dict(finalize_by_runs=True)),
    ("mh_by_runs", [("m", "CG"), ("h", "CG")], [0.8, 0.7], dict(), dict(finalize_by_runs=True)),
    ("haplotypes", [("m", "CG")], [0.8], dict(haplotypes=True), dict()),
    ("insertions_haplotypes", [("m", "CG")], [0.8], dict(insertions=True, haplotypes=True), dict()),
    ("fused", [("m", "CG"), ("h", "CG")], [0.8, 0.7], dict(), dict(force_fused=True)),
]


@pytest.mark.parametrize("name,mods,th,okw,ekw", CASES, ids=[c[0] for c in CASES])
def test_library_rows_in_reference_contig_order(name, mods, th, okw, ekw, genome):
    """Rows in output order, element for element (both sides put ties in the canonical order)."""
    gen = dict(haplotypes=True, long_insertions=True) if okw.get("haplotypes") else {}
    bs = _batches(genome, **gen)
    assert any(len(set(b["reads"]["tid"].tolist())) > 1 for b in bs)     # a batch that runs across a contig boundary
    eng = _engine(genome, mods, th, **okw, **ekw)
    tk = []
    for b in bs:
        tk.append(eng.submit(b))
        if len(tk) >= 3:
            eng.wait(tk.pop(0))
    for t in tk:
        eng.wait(t)
    got = eng.finalize()
    eng.close()
    want = _oracle(genome, bs, mods, th, **okw).rows()
    assert len(want) > 10000
    assert [NAMES[t] for t in dict.fromkeys(want["tid"].tolist())] == ["chr1", "chr10", "chr2", "chrX"]
    assert _key_rows(got, "ins_offset") == _key_rows(want, "ins_off")


def test_read_on_contig_missing_from_fasta_fails_like_reference(genome):
    """src/mod.c:793: a read whose contig the FASTA lacks is an error, not a skipped read."""
    import minimod_amd
    from minimod_amd import synth
    b = synth.batch(genome[0][:LENS[4]], 0, 5, seed=1, n_reads_total=5, tid=4, median_len=500.0, max_len=2000.0)
    eng = _engine(genome, [("m", "CG")], [0.8])
    with pytest.raises(minimod_amd.engine.MinimodHipError) as ei:
        eng.process(b)
    assert ei.value.code == 12 and ei.value.read == 0
    eng.close()


def _write_inputs(tmp_path, genome, batches, fasta_order=(3, 0, 2, 1)):
    from minimod_amd import synth
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    synth.write_bam(bam, list(zip(NAMES, LENS)), batches)
    # FASTA in another order than the BAM header, plus a contig the BAM does not know
    extra = ("chrUn_decoy", synth.reference(999, 70000))
    synth.write_fasta_multi(fa, [(NAMES[i], genome[i]) for i in fasta_order[:2]] + [extra] + [(NAMES[i], genome[i]) for i in fasta_order[2:]])
    return bam, fa


@pytest.mark.parametrize("c,m,flags", [("m[CG]", "0.8", ["-b"]), ("m[CG]", "0.8", []), ("m[CG],h[CG]", "0.8,0.7", ["-b"])],
                         ids=["bedmethyl", "tsv", "bedmethyl_mh"])
def test_cli_multi_contig_matches_oracle(c, m, flags, genome, tmp_path):
    bs = _batches(genome)
    bam, fa = _write_inputs(tmp_path, genome, bs)
    mods = O.parse_mod_codes(c)
    th = O.parse_mod_threshes(m, len(mods))
    orc = _oracle(genome, bs, mods, th)
    want = O.format_rows(orc.rows(), NAMES, orc.code_names(), bedmethyl="-b" in flags)
    outs, replayed = [], []
    for extra in (["-K", "97", "-t", "3"], ["-K", "4096", "-B", "100M", "-t", "8"]):
        # the oracle puts rows that tie on (contig, start) in the canonical order: so does the CLI with --canonical-order
        r = subprocess.run([BIN, "freq", "--canonical-order", "-c", c, "-m", m] + flags + extra + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(r.stdout.decode())
        r = subprocess.run([BIN, "freq", "-c", c, "-m", m] + flags + extra + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        replayed.append(r.stdout.decode())
    assert len(want) > 100000 and outs[0] == want and outs[1] == want
    # the reference's order (replayed): the same rows, the same bytes whatever the batching; without ties the same as above
    assert replayed[0] == replayed[1] and sorted(replayed[0].splitlines()) == sorted(want.splitlines())
    if "," not in c:
        assert replayed[0] == want
    else:
        assert replayed[0] != want      # two codes on every CpG: the hash order differs from the canonical one somewhere


def test_cli_view_multi_contig_matches_oracle(genome, tmp_path):
    bs = _batches(genome)
    bam, fa = _write_inputs(tmp_path, genome, bs)
    r = subprocess.run([BIN, "view", "-c", "m[CG]", "-K", "200", "-t", "4", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    contigs = {n: g for n, g in zip(NAMES, genome) if g is not None}
    rows, qn, names, codes = O.view(bam, contigs, c="m[CG]", K=200, threads=4)
    want = O.format_view(rows, qn, names, codes)
    assert len(want) > 100000 and r.stdout.decode() == want


def test_contig_sharded_handles_union_equals_unsharded(genome):
    """Two handles own different contig sets (whole-contig intervals, as ranks of a contig-sharded run would); each gets the
    reads of its contigs; rows concatenated in contig order equal the unsharded oracle."""
    bs = _batches(genome, batch_reads=100000)   # one batch: the whole read set
    whole = bs[0]
    owners = {0: [0, 3], 1: [1, 2]}
    got = []
    for rank, tids in owners.items():
        from minimod_amd import synth
        sel = np.isin(whole["reads"]["tid"], tids)
        part = dict(whole)
        part["reads"] = np.ascontiguousarray(whole["reads"][sel])
        eng = _engine(genome, [("m", "CG")], [0.8], intervals=[(t, 0, LENS[t], 0) for t in tids])
        for sub in synth.split(part, [min(300, len(part["reads"]) - i) for i in range(0, len(part["reads"]), 300)]):
            eng.process(sub)
        got.append(eng.finalize())
        eng.close()
    want = _oracle(genome, bs, [("m", "CG")], [0.8]).rows()
    allrows = np.concatenate(got)
    rank_of = {n: i for i, n in enumerate(sorted(NAMES))}
    order = np.argsort(np.array([rank_of[NAMES[t]] for t in allrows["tid"]]), kind="stable")   # per-handle rows are already in (contig, pos) order
    allrows = allrows[order]
    assert _key_rows(allrows, "ins_offset") == _key_rows(want, "ins_off")


def test_downscaled_whole_genome_shape(tmp_path):
    """C4 scaled by 1/2000: 24 contigs named chr1..chr22, chrX, chrY with hg38's length ratios, 30x of ONT-shape reads, one
    coordinate-sorted BAM through the CLI with the reference's production flags (-K 4092 -B 100M, test/test_ext.sh:63):
    bedmethyl byte-identical to the oracle's."""
    from minimod_amd import synth
    hg38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
            133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895,
            57227415]
    names = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
    lens = [max(20000, L // 2000) for L in hg38]
    refs = [synth.reference(500 + i, L) for i, L in enumerate(lens)]
    nreads = [max(4, int(30 * L / 9000)) for L in lens]
    bs = synth.multi_contig(refs, nreads, 1024, seed=77, median_len=7000.0, max_len=60000.0)
    bam, fa = str(tmp_path / "wg.bam"), str(tmp_path / "wg.fa")
    synth.write_bam(bam, list(zip(names, lens)), bs)
    synth.write_fasta_multi(fa, list(zip(names, refs)))
    r = subprocess.run([BIN, "freq", "-c", "m[CG]", "-K", "4092", "-B", "100M", "-b", "-t", "8", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    orc = O.Oracle([("m", "CG")], [0.8], names)
    for n, rf in zip(names, refs):
        orc.add_contig(n, rf)
    for b in bs:
        orc.process(b, threads=8)
    rows = orc.rows()
    want = O.format_rows(rows, names, orc.code_names(), bedmethyl=True)
    first_seen = [names[t] for t in dict.fromkeys(rows["tid"].tolist())]
    assert first_seen == sorted(names) and first_seen[:3] == ["chr1", "chr10", "chr11"]
    assert len(rows) > 20000 and r.stdout.decode() == want


@pytest.mark.parametrize("flags", [["-b"], [], ["--haplotypes", "--insertions"]], ids=["bedmethyl", "tsv", "hap_ins"])
def test_cli_devices_shares_of_the_genome_equal_a_single_run(flags, genome, tmp_path):
    """`minimod freq --devices a,b,c`: one worker process per listed GPU (the test lists the one GPU there is three times),
    the genome cut into contiguous shares at 64 kb-aligned positions (cuts fall inside contigs here), every worker reading
    its share of the BAM from the virtual offset the .bai gives, rows merged by the parent: the same bytes as one run."""
    from minimod_amd import synth
    gen = dict(haplotypes=True, long_insertions=True) if "--haplotypes" in flags else {}
    bs = _batches(genome, **gen)
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    synth.write_bam(bam, list(zip(NAMES, LENS)), bs, index=True)
    synth.write_fasta_multi(fa, [(n, g) for n, g in zip(NAMES, genome) if g is not None])
    assert os.path.getsize(bam + ".bai") > 1000
    base = [BIN, "freq", "--canonical-order", "-c", "m[CG],h[CG]", "-m", "0.8,0.7", "-K", "200", "-t", "6"] + flags
    one = subprocess.run(base + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    for devs in ("0,0", "0,0,0"):
        many = subprocess.run(base + ["--devices", devs, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert many.returncode == 0, many.stderr.decode()[-3000:]
        assert len(one.stdout) > 100000 and many.stdout == one.stdout
        assert b"total processed entries: 1100" in many.stderr and b"devices: %d" % len(devs.split(",")) in many.stderr
        # the halo slabs go from GPU to GPU through HIP IPC handles (two workers on one GPU exercise it too); through host memory
        # and the socket when that is switched off: the same bytes either way
        assert b"through a HIP IPC handle" in many.stderr, many.stderr.decode()[-1500:]
    old = subprocess.run(base + ["--devices", "0,0", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, MM_NO_IPC_SLABS="1"))
    assert old.returncode == 0 and old.stdout == one.stdout and b"through a HIP IPC handle" not in old.stderr


@pytest.mark.parametrize("flags", [[], ["--haplotypes", "--insertions"], ["-b"]], ids=["tsv", "hap_ins", "bedmethyl"])
def test_cli_devices_print_tied_rows_in_the_reference_order(flags, genome, tmp_path):
    """Without --canonical-order a --devices run prints rows that tie on (contig, start) in the order the reference's hash table
    and sort leave them in, like a single run: every worker replays its own reads' keys, the parent strings the workers'
    first-insertion sequences together in file order (src/mod.c:743-774) and orders the rows.  Same bytes as one run."""
    from minimod_amd import synth
    gen = dict(haplotypes=True, long_insertions=True) if "--haplotypes" in flags else {}
    bs = _batches(genome, **gen)
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    synth.write_bam(bam, list(zip(NAMES, LENS)), bs, index=True)
    synth.write_fasta_multi(fa, [(n, g) for n, g in zip(NAMES, genome) if g is not None])
    base = [BIN, "freq", "-c", "m[CG],h[CG]", "-m", "0.8,0.7", "-K", "200", "-t", "6"] + flags
    one = subprocess.run(base + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert one.returncode == 0 and b"Row order replay" in one.stderr, one.stderr.decode()[-2000:]
    canon = subprocess.run(base + ["--canonical-order", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert canon.returncode == 0 and sorted(canon.stdout.splitlines()) == sorted(one.stdout.splitlines())
    if "-b" not in flags:
        assert canon.stdout != one.stdout          # (the two orders do differ on this input: the test means something)
    for devs in ("0,0", "0,0,0"):
        many = subprocess.run(base + ["--devices", devs, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert many.returncode == 0, many.stderr.decode()[-3000:]
        assert len(one.stdout) > 100000 and many.stdout == one.stdout, many.stderr.decode()[-1500:]


def test_cli_view_devices_equal_a_single_run(genome, tmp_path):
    """`minimod view --devices a,b,c`: every read is one worker's (the one its start lies in); the workers' rows one after the
    other are the single run's rows."""
    from minimod_amd import synth
    bs = _batches(genome)
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    synth.write_bam(bam, list(zip(NAMES, LENS)), bs, index=True)
    synth.write_fasta_multi(fa, [(n, g) for n, g in zip(NAMES, genome) if g is not None])
    base = [BIN, "view", "-c", "m[CG],h[CG]", "-K", "200", "-t", "6"]
    one = subprocess.run(base + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    for devs in ("0,0", "0,0,0"):
        many = subprocess.run(base + ["--devices", devs, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert many.returncode == 0, many.stderr.decode()[-3000:]
        assert len(one.stdout) > 100000 and many.stdout == one.stdout
    # round 5: view's workers read their shares with the device-side reader too (the read names come with the batches); small groups:
    # several batches a worker, records carried from group to group
    for devs, env in (("0,0", {}), ("0,0,0", {"MM_INGEST_MAX_BLOCKS": "8", "MM_INGEST_TARGET_BASES": "200000"})):
        many = subprocess.run(base + ["--gpu-ingest", "--devices", devs, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, **env))
        assert many.returncode == 0, many.stderr.decode()[-3000:]
        assert b"[gpu-ingest]" in many.stderr and many.stdout == one.stdout
    single = subprocess.run(base + ["--gpu-ingest", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert single.returncode == 0 and b"[gpu-ingest]" in single.stderr and single.stdout == one.stdout


def test_cli_devices_rejects_a_malformed_list(genome, tmp_path):
    from minimod_amd import synth
    bs = _batches(genome)
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    synth.write_bam(bam, list(zip(NAMES, LENS)), bs, index=True)
    synth.write_fasta_multi(fa, [(n, g) for n, g in zip(NAMES, genome) if g is not None])
    for bad in ("0,,1", "a,b", "0,-1", "0;1"):
        r = subprocess.run([BIN, "freq", "--devices", bad, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 1 and b"--devices takes GPU ordinals" in r.stderr, (bad, r.stderr.decode()[-500:])


def test_cli_devices_needs_the_index(genome, tmp_path):
    from minimod_amd import synth
    bs = _batches(genome)
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    synth.write_bam(bam, list(zip(NAMES, LENS)), bs)
    synth.write_fasta_multi(fa, [(n, g) for n, g in zip(NAMES, genome) if g is not None])
    r = subprocess.run([BIN, "freq", "--devices", "0,0", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 1 and b"Could not read the index" in r.stderr


@pytest.mark.parametrize("cfg", [(["-c", "m[CG]", "-m", "0.8"], True), (["-b", "-c", "m[CG]"], True), (["-c", "m[CG],h[CG]", "-m", "0.8,0.7"], False), (["-c", "m[CG]", "--haplotypes", "--insertions"], False)],
                         ids=["m", "bedmethyl", "m_h", "hap_ins"])
def test_cli_region_equals_the_full_run_restricted_to_the_region(cfg, genome, tmp_path):
    """`minimod freq --region chr:from-to` (the reference's own region code is commented out, src/minimod.c:92-130):
    the reads that can reach the region come through the .bai's linear index, the dense counters cover the region alone, rows outside
    are dropped -- what is printed is the FULL run's rows of those positions (counts included: every read over the region is in), in
    the full run's order where rows cannot tie; with the host threads and with the device-side reader."""
    from minimod_amd import synth
    flags, exact = cfg
    gen = dict(haplotypes=True, long_insertions=True) if "--haplotypes" in flags else {}
    bs = _batches(genome, **gen)
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    synth.write_bam(bam, list(zip(NAMES, LENS)), bs, index=True)
    synth.write_fasta_multi(fa, [(n, g) for n, g in zip(NAMES, genome) if g is not None])
    base = [BIN, "freq", "-K", "200", "-t", "4"] + flags
    full = subprocess.run(base + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert full.returncode == 0, full.stderr.decode()[-2000:]
    lines = full.stdout.decode().splitlines(True)
    header = [l for l in lines if l.startswith("contig\t")]
    body = [l for l in lines if not l.startswith("contig\t")]
    for region, name, lo, hi in (("chr2:300,001-500000", "chr2", 300000, 500000), ("chr10", "chr10", 0, LENS[2]), ("chrX:1-70000", "chrX", 0, 70000),
                                 ("chr1:1040000-", "chr1", 1039999, LENS[0]), ("chr2:777777-777777", "chr2", 777776, 777777)):
        want = [l for l in body if l.split("\t")[0] == name and lo <= int(l.split("\t")[1]) < hi]
        for extra in ([], ["--gpu-ingest"]):
            r = subprocess.run(base + extra + ["--region", region, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            got = r.stdout.decode().splitlines(True)
            assert got[:len(header)] == header
            got = got[len(header):]
            if exact:
                assert got == want, (region, extra)
            else:
                assert sorted(got) == sorted(want), (region, extra)
            assert len(want) > 0 or region.endswith("777777")
            assert b"--region" in r.stderr and (b"[gpu-ingest]" in r.stderr) == bool(extra)
    bad = subprocess.run(base + ["--region", "chrNone:1-5", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert bad.returncode == 1 and b"no contig chrNone" in bad.stderr
    os.remove(bam + ".bai")
    noidx = subprocess.run(base + ["--region", "chr2", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert noidx.returncode == 1 and b"index" in noidx.stderr


def test_cli_eight_workers_on_one_gpu_equal_a_single_run(genome, tmp_path):
    """`--devices 0,0,0,0,0,0,0,0`: the eight workers of a node, all on the one GPU there is -- eight processes at once on a device (the condition
    round 5's wrong site index needed), seven halo slabs through HIP IPC handles, eight device readers on one host.  The same bytes as one run, with
    rows that tie in the reference's order as well."""
    from minimod_amd import synth
    bs = _batches(genome)
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    synth.write_bam(bam, list(zip(NAMES, LENS)), bs, index=True)
    synth.write_fasta_multi(fa, [(n, g) for n, g in zip(NAMES, genome) if g is not None])
    for order in (["--canonical-order"], []):
        base = [BIN, "freq"] + order + ["-b", "-c", "m[CG],h[CG]", "-m", "0.8,0.7", "-K", "200", "-t", "8"]
        one = subprocess.run(base + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert one.returncode == 0, one.stderr.decode()[-2000:]
        for attempt in range(3):
            many = subprocess.run(base + ["--devices", "0,0,0,0,0,0,0,0", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert many.returncode == 0, many.stderr.decode()[-3000:]
            assert len(one.stdout) > 100000 and many.stdout == one.stdout
            assert b"devices: 8" in many.stderr and b"failed its check" not in many.stderr
            # every cut that falls inside a contig hands its slab over through a HIP IPC handle
            assert many.stderr.count(b"through a HIP IPC handle") >= 3 and b"through host memory" not in many.stderr   # (a share without calls near its cut sends no slab), many.stderr.decode()[-1500:]
