"""End-to-end `minimod freq` (C host + HIP library) against the reference's golden files.  GPU only."""
import os
import subprocess

import numpy as np
import pytest

from tests.cases import CLI_SORTED_ONLY, GOLDEN, GOLDEN_CASES

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "minimod_amd", "bin", "minimod")


def _write_fasta(path, name, seq):
    with open(path, "wb") as f:
        f.write(b">" + name.encode() + b" pseudo-reference\n")
        a = np.frombuffer(seq, dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
        n = len(a) // 60 * 60
        body = np.empty((n // 60, 61), dtype=np.uint8)
        body[:, :60] = a[:n].reshape(-1, 60)
        body[:, 60] = 10
        f.write(body.tobytes())
        if n < len(a):
            f.write(a[n:].tobytes() + b"\n")


@pytest.fixture(scope="session")
def fastas(tmp_path_factory, chr22, chr1):
    import minimod_amd
    minimod_amd.build_all()
    d = tmp_path_factory.mktemp("fa")
    out = {}
    for key, ctg in (("chr22", chr22), ("chr1", chr1)):
        (name, seq), = ctg.items()
        p = str(d / (key + ".fa"))
        _write_fasta(p, name, seq)
        out[key] = p
    return out


def _args(kw):
    a = []
    if "c" in kw:
        a += ["-c", kw["c"]]
    if "m" in kw:
        a += ["-m", kw["m"]]
    if "K" in kw:
        a += ["-K", str(kw["K"])]
    if kw.get("insertions"):
        a.append("--insertions")
    if kw.get("haplotypes"):
        a.append("--haplotypes")
    return a


@pytest.mark.parametrize("exp,bam,ctg,kw,exact", GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def test_cli_matches_reference_golden(exp, bam, ctg, kw, exact, fastas, tmp_path):
    cmd = [BIN, "freq"] + _args(kw) + (["-b"] if exp.endswith("bedmethyl") else []) + [fastas[ctg], os.path.join(GOLDEN, "data", bam)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    want = open(os.path.join(GOLDEN, "expected", exp)).read()
    got = r.stdout.decode()
    # byte for byte, ties included: the CLI replays the order the reference's hash table and unstable sort leave them in
    if exp in CLI_SORTED_ONLY:
        assert sorted(got.splitlines()) == sorted(want.splitlines())
    else:
        assert got == want
    if not exact:   # --canonical-order: the same rows, ties in the fixed order; never the replay
        r2 = subprocess.run(cmd[:2] + ["--canonical-order"] + cmd[2:], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r2.returncode == 0 and sorted(r2.stdout.decode().splitlines()) == sorted(want.splitlines())
        assert b"Row order replay" not in r2.stderr and b"Row order replay" in r.stderr


@pytest.mark.parametrize("exp,bam,ctg,kw,exact", GOLDEN_CASES, ids=["devfmt-" + c[0] for c in GOLDEN_CASES])
def test_cli_device_formatter_matches_reference_golden(exp, bam, ctg, kw, exact, fastas):
    """the same goldens with the rows' text made on the device (MINIMOD_FMT=device: the default from 65 536 rows on; mm_fmt_rows,
    SURVEY 8(f) row 3): byte for byte, the tie order replayed on the device in front of it"""
    cmd = [BIN, "freq"] + _args(kw) + (["-b"] if exp.endswith("bedmethyl") else []) + [fastas[ctg], os.path.join(GOLDEN, "data", bam)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, MINIMOD_FMT="device"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"rows formatted on the device" in r.stderr
    want = open(os.path.join(GOLDEN, "expected", exp)).read()
    if exp in CLI_SORTED_ONLY:
        assert sorted(r.stdout.decode().splitlines()) == sorted(want.splitlines())
    else:
        assert r.stdout.decode() == want
    # (a run without ties leaves its rows in GPU memory for the formatter, mm_freq_finalize_device; MM_ROWS_TO_HOST=1: through the host as before)
    r2 = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, MINIMOD_FMT="device", MM_ROWS_TO_HOST="1"))
    assert r2.returncode == 0 and r2.stdout == r.stdout, r2.stderr.decode()[-2000:]


def test_cli_tie_order_does_not_depend_on_batching(fastas):
    """The replayed order is a function of the reads in file order only (reference invariance): the
    same bytes for -K 1, -K 7 and -K 4096 with different thread counts, on the goldens that tie."""
    for exp, bam, ctg, kw, _ in [c for c in GOLDEN_CASES if c[0] in ("test5a.tsv", "test8.tsv", "test5c.tsv")]:
        want = open(os.path.join(GOLDEN, "expected", exp)).read()
        for extra in (["-K", "1", "-t", "2"], ["-K", "7", "-t", "8"], ["-K", "4096", "-B", "100M", "-t", "3"]):
            r = subprocess.run([BIN, "freq"] + _args(kw) + extra + [fastas[ctg], os.path.join(GOLDEN, "data", bam)], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=300)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            assert r.stdout.decode() == want, (exp, extra)


def test_cli_output_file_and_batch_invariance(fastas, tmp_path):
    bam = os.path.join(GOLDEN, "data", "example-ont.bam")
    outs = []
    for i, extra in enumerate((["-K", "1"], ["-K", "4096", "-B", "100M", "-t", "4"], ["-B", "50K"])):
        o = str(tmp_path / ("o%d.tsv" % i))
        r = subprocess.run([BIN, "freq", "-m", "0.8", "-o", o] + extra + [fastas["chr22"], bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0 and r.stdout == b""
        outs.append(open(o).read())
    assert outs[0] == outs[1] == outs[2] == open(os.path.join(GOLDEN, "expected", "test7.tsv")).read()


def test_cli_hard_clip_exits_like_reference(fastas, tmp_path):
    r = subprocess.run([BIN, "freq", fastas["chr22"], os.path.join(GOLDEN, "data", "dna_5mCG_5hmCG_mm_with_secondary_chr22.bam")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    # the reference dies on the hard-clipped supplementary alignments of this file unless they are filtered
    if r.returncode != 0:
        assert b"Hard clipping" in r.stderr
    r2 = subprocess.run([BIN, "freq", "--skip-supplementary", fastas["chr22"],
                         os.path.join(GOLDEN, "data", "dna_5mCG_5hmCG_mm_with_secondary_chr22.bam")],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r2.returncode == 0 and len(r2.stdout) > 1000


def test_cli_missing_args_and_version():
    r = subprocess.run([BIN, "freq"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Usage: minimod freq ref.fa reads.bam" in r.stderr
    r = subprocess.run([BIN, "freq", "-h"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"Usage: minimod freq ref.fa reads.bam" in r.stdout
    r = subprocess.run([BIN, "--version"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout.startswith(b"minimod ")


def test_cli_on_synthetic_bam_matches_oracle(tmp_path):
    """20 Mbases of synthetic ONT-shape reads as a real BGZF BAM + FASTA through the CLI (-K 512, several batches,
    filter fodder in the file): bedmethyl byte-identical to the oracle's."""
    from minimod_amd import synth
    from oracle import oracle as O
    ref = synth.reference(13, 4 << 20)
    bs = [synth.batch(ref, i * 350, 350, seed=3, n_reads_total=1400) for i in range(4)]
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    synth.write_bam(bam, [("chrS", len(ref))], bs)
    synth.write_fasta(fa, "chrS", ref)
    r = subprocess.run([BIN, "freq", "-b", "-c", "m[CG]", "-m", "0.8", "-K", "512", "-B", "100M", "-t", "4", fa, bam],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    orc = O.Oracle([("m", "CG")], [0.8], ["chrS"])
    orc.add_contig("chrS", ref)
    for b in bs:
        orc.process(b, threads=4)
    want = O.format_rows(orc.rows(), ["chrS"], orc.code_names(), bedmethyl=True)
    assert len(want) > 100000 and r.stdout.decode() == want
    assert b"total processed entries: 1400" in r.stderr


# ---- view (reference test/test.sh:66-111,186-247)
from tests.cases import VIEW_CASES  # noqa: E402


@pytest.mark.parametrize("exp,bam,ctg,kw,exact", VIEW_CASES, ids=["view-" + c[0] for c in VIEW_CASES])
def test_cli_view_matches_reference_golden(exp, bam, ctg, kw, exact, fastas, tmp_path):
    cmd = [BIN, "view"] + _args(kw) + [fastas[ctg], os.path.join(GOLDEN, "data", bam)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    want = open(os.path.join(GOLDEN, "expected", exp)).read()
    got = r.stdout.decode()
    if exact:
        assert got == want
    else:
        assert sorted(got.splitlines()) == sorted(want.splitlines())


def test_cli_view_output_file_and_batch_invariance(fastas, tmp_path):
    """Test 13 of the reference (view -o) and: the rows do not depend on -K / -B / -t."""
    bam = os.path.join(GOLDEN, "data", "example-ont.bam")
    outs = []
    for i, extra in enumerate((["-K", "1"], ["-K", "4096", "-B", "100M", "-t", "4"], ["-B", "50K"])):
        o = str(tmp_path / ("v%d.tsv" % i))
        r = subprocess.run([BIN, "view", "-o", o] + extra + [fastas["chr22"], bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0 and r.stdout == b"", r.stderr.decode()[-2000:]
        outs.append(open(o).read())
    assert outs[0] == outs[1] == outs[2] == open(os.path.join(GOLDEN, "expected", "test10.tsv")).read()


def test_cli_view_rejects_freq_only_options(fastas):
    bam = os.path.join(GOLDEN, "data", "example-ont.bam")
    for opt in (["-b"], ["-m", "0.8"]):   # not in view's option table (src/view_main.c:46-63,168): help + failure
        r = subprocess.run([BIN, "view"] + opt + [fastas["chr22"], bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1 and b"Usage: minimod view ref.fa reads.bam" in r.stderr


def test_cli_view_on_synthetic_bam_matches_oracle(tmp_path):
    """Synthetic ONT-shape reads as a real BGZF BAM + FASTA through `minimod view` (several batches, both strands,
    insertions): byte-identical to the oracle's view of the same file."""
    from minimod_amd import synth
    from oracle import oracle as O
    ref = synth.reference(13, 1 << 20)
    bs = [synth.batch(ref, i * 100, 100, seed=5, n_reads_total=300) for i in range(3)]
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    synth.write_bam(bam, [("chrS", len(ref))], bs)
    synth.write_fasta(fa, "chrS", ref)
    for extra, kw in (([], dict()), (["--insertions"], dict(insertions=True))):
        r = subprocess.run([BIN, "view", "-c", "m[CG]", "-K", "128", "-t", "4"] + extra + [fa, bam],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        rows, qn, names, codes = O.view(bam, {"chrS": ref}, c="m[CG]", K=128, threads=4, **kw)
        want = O.format_view(rows, qn, names, codes, **kw)
        assert len(want) > 100000 and r.stdout.decode() == want


def test_cli_on_bam_without_records(tmp_path):
    """A BAM that holds only its header: both subtools print their header line (bedmethyl: nothing) and exit 0."""
    from minimod_amd import synth
    ref = synth.reference(3, 1 << 16)
    bam, fa = str(tmp_path / "e.bam"), str(tmp_path / "e.fa")
    synth.write_bam(bam, [("chrS", len(ref))], [])
    synth.write_fasta(fa, "chrS", ref)
    r = subprocess.run([BIN, "freq", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0 and r.stdout.decode() == "contig\tstart\tend\tstrand\tn_called\tn_mod\tfreq\tmod_code\n", r.stderr.decode()[-1000:]
    r = subprocess.run([BIN, "freq", "-b", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0 and r.stdout == b""
    r = subprocess.run([BIN, "view", "--insertions", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0 and r.stdout.decode() == "ref_contig\tref_pos\tstrand\tread_id\tread_pos\tmod_code\tmod_prob\tins_offset\n"


def test_cli_view_hard_clip_exits_like_reference(fastas):
    r = subprocess.run([BIN, "view", "--allow-secondary", fastas["chr22"], os.path.join(GOLDEN, "data", "dna_5mCG_5hmCG_mm_with_secondary_chr22.bam")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    if r.returncode != 0:
        assert r.returncode == 1 and b"Hard clipping" in r.stderr


# ---- --gpu-ingest: the decoded BAM stays in GPU memory (include/minimod_ingest.h, csrc/host/devloader.c)
INGEST_CASES = [c for c in GOLDEN_CASES if c[4]]   # the runs whose rows cannot tie: the device loader's (the others fall back to the host threads)


@pytest.mark.parametrize("exp,bam,ctg,kw,exact", INGEST_CASES, ids=["ingest-" + c[0] for c in INGEST_CASES])
def test_cli_gpu_ingest_matches_reference_golden(exp, bam, ctg, kw, exact, fastas):
    cmd = [BIN, "freq", "--gpu-ingest"] + _args(kw) + (["-b"] if exp.endswith("bedmethyl") else []) + [fastas[ctg], os.path.join(GOLDEN, "data", bam)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"[gpu-ingest]" in r.stderr
    assert r.stdout.decode() == open(os.path.join(GOLDEN, "expected", exp)).read()
    # the totals the reference prints are the host loader's
    r0 = subprocess.run([c for c in cmd if c != "--gpu-ingest"] + ["--no-gpu-ingest"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    pick = lambda err: [l.split("] ", 1)[1] for l in err.decode().splitlines() if "] total " in l]
    assert pick(r.stderr) == pick(r0.stderr) and len(pick(r.stderr)) == 7


TIED_CASES = [c for c in GOLDEN_CASES if not c[4] and c[0] not in CLI_SORTED_ONLY]   # test5a, 5b, 5c, 8, 12: rows tie, the reference's order is replayed


@pytest.mark.parametrize("exp,bam,ctg,kw,exact", TIED_CASES, ids=["tied-" + c[0] for c in TIED_CASES])
def test_cli_tied_runs_replay_on_the_device_and_take_the_device_reader(exp, bam, ctg, kw, exact, fastas):
    """Round 5: the reference's tie order comes from the device-side replay (include/minimod_tie.h) by default -- with
    the host threads reading, with the device-side reader (--gpu-ingest no longer falls back for tied runs), and for small groups of BGZF
    blocks (several launches, tails carried over); --host-replay is round 4's serial restatement, the checker: the same bytes."""
    want = open(os.path.join(GOLDEN, "expected", exp)).read()
    base = [BIN, "freq"] + _args(kw)
    tail = [fastas[ctg], os.path.join(GOLDEN, "data", bam)]
    for extra, env, marks in (([], {}, [b"on the device"]),
                              (["--gpu-ingest"], {}, [b"on the device", b"[gpu-ingest]"]),
                              (["--gpu-ingest"], {"MM_INGEST_MAX_BLOCKS": "8", "MM_INGEST_TARGET_BASES": "200000"}, [b"on the device", b"[gpu-ingest]"]),
                              (["--host-replay"], {}, [b"on the host"]),
                              (["--gpu-ingest", "--host-replay"], {}, [b"on the host"]),
                              (["-K", "3"], {"MINIMOD_HOST_REPLAY": "1"}, [b"on the host"])):
        r = subprocess.run(base + extra + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        for m in marks:
            assert m in r.stderr, (extra, m, r.stderr.decode()[-1500:])
        if b"on the host" in marks[0]:
            assert b"[gpu-ingest]" not in r.stderr   # the host replay works from host batches
        assert r.stdout.decode() == want, (exp, extra, env)


def test_cli_goes_on_with_the_host_reader_when_the_device_reader_cannot_take_the_file(tmp_path):
    """--gpu-ingest switches itself on for big files, so a file the device reader cannot take (here: records that straddle
    its groups and are longer than its head room, forced through the environment; htslib's own files end their blocks on record boundaries,
    the synthetic writer does not) must not fail the run -- nothing has been counted when it gives up on its first group, the host reader
    takes over in the same process: a warning, the same bytes and totals as --no-gpu-ingest, tied runs (device replay) included"""
    from minimod_amd import synth
    ref = synth.reference(13, 4 << 20)
    bs = [synth.batch(ref, i * 350, 350, seed=3, n_reads_total=700) for i in range(2)]
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    synth.write_bam(bam, [("chrS", len(ref))], bs)
    synth.write_fasta(fa, "chrS", ref)
    for flags in (["-b", "-c", "m[CG]", "-m", "0.8"], ["-c", "m[CG],h[CG]", "-m", "0.8,0.7"]):
        cmd = [BIN, "freq", "-t", "4"] + flags
        r = subprocess.run(cmd + ["--gpu-ingest", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, MM_INGEST_HEAD_ROOM="256", MM_INGEST_MAX_BLOCKS="8"))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert b"the host threads read the file" in r.stderr and b"gave up" in r.stderr, r.stderr.decode()[-2000:]
        r0 = subprocess.run(cmd + ["--no-gpu-ingest", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r0.returncode == 0 and len(r0.stdout) > 10000 and r.stdout == r0.stdout
        pick = lambda err: [l.split("] ", 1)[1] for l in err.decode().splitlines() if "] total " in l]
        assert pick(r.stderr) == pick(r0.stderr) and len(pick(r.stderr)) == 7


def test_cli_gpu_ingest_on_synthetic_bam(tmp_path):
    """the synthetic ONT-shape file of test_cli_on_synthetic_bam_matches_oracle (filter fodder, records that straddle BGZF blocks and
    groups): the same bytes with the device loader, with small groups forced through the environment, and a hard-clipped read
    reported as the host path reports it"""
    from minimod_amd import synth
    from oracle import oracle as O
    ref = synth.reference(13, 4 << 20)
    bs = [synth.batch(ref, i * 350, 350, seed=3, n_reads_total=1400) for i in range(4)]
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    synth.write_bam(bam, [("chrS", len(ref))], bs)
    synth.write_fasta(fa, "chrS", ref)
    orc = O.Oracle([("m", "CG")], [0.8], ["chrS"])
    orc.add_contig("chrS", ref)
    for b in bs:
        orc.process(b, threads=4)
    want = O.format_rows(orc.rows(), ["chrS"], orc.code_names(), bedmethyl=True)
    for env in ({}, {"MM_INGEST_MAX_BLOCKS": "16", "MM_INGEST_TARGET_BASES": "3000000"}):
        r = subprocess.run([BIN, "freq", "--gpu-ingest", "-b", "-c", "m[CG]", "-m", "0.8", "-t", "4", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert r.stdout.decode() == want
        assert b"total processed entries: 1400" in r.stderr and b"[gpu-ingest]" in r.stderr
    # a truncated file fails the run, as it does with the host reader
    cut = str(tmp_path / "cut.bam")
    raw = open(bam, "rb").read()
    open(cut, "wb").write(raw[:len(raw) * 2 // 3])
    r = subprocess.run([BIN, "freq", "--gpu-ingest", "-b", fa, cut], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 1 and b"Truncated or corrupt BAM file" in r.stderr


# ---- `minimod summary --gpu`: the census kernel (include/minimod_summary.h)
def test_cli_summary_census_on_the_device_matches_reference_goldens():
    """the reference's five summary goldens (test/test.sh:252-256,494-503; plain diff there: the order of a read's keys is the slot order of
    its khash) through k_sum_reads -- a thread per read walks the MM groups and fills the read's table as khash would -- for two batchings,
    and the same bytes as the host walk on a file with every kind of group (ChEBI codes, multi-letter codes, groups without calls)"""
    from tests.test_host_cpu import SUMMARY_CASES
    for exp, bam, extra in SUMMARY_CASES:
        for more in ([], ["-K", "7", "-t", "3"]):
            r = subprocess.run([BIN, "summary", "--gpu"] + extra + more + [os.path.join(GOLDEN, "data", bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            assert b"census on the device" in r.stderr
            assert r.stdout.decode() == open(os.path.join(GOLDEN, "expected", exp)).read(), (exp, more)
    for bam in sorted(f for f in os.listdir(os.path.join(GOLDEN, "data")) if f.endswith(".bam")):
        a = subprocess.run([BIN, "summary", os.path.join(GOLDEN, "data", bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        b = subprocess.run([BIN, "summary", "--gpu", os.path.join(GOLDEN, "data", bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert a.returncode == b.returncode and a.stdout == b.stdout, bam


# ---- round 5: view and -c '*' take the device-side reader too ("device ingestion everywhere")
@pytest.mark.parametrize("exp,bam,ctg,kw,exact", VIEW_CASES, ids=["ingest-view-" + c[0] for c in VIEW_CASES])
def test_cli_view_with_the_device_reader_matches_reference_golden(exp, bam, ctg, kw, exact, fastas):
    """`minimod view --gpu-ingest`: the read names come with the device batches (mm_ingest_arena_names), a wildcard run's codes from the
    census kernel (mm_ingest_batch_codes) -- the reference's goldens, and the bytes of the host threads' run, also with groups of 8 blocks
    (several batches, records carried from group to group)"""
    tail = [fastas[ctg], os.path.join(GOLDEN, "data", bam)]
    want = open(os.path.join(GOLDEN, "expected", exp)).read()
    r0 = subprocess.run([BIN, "view", "--no-gpu-ingest"] + _args(kw) + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r0.returncode == 0 and b"[gpu-ingest]" not in r0.stderr
    for env in ({}, {"MM_INGEST_MAX_BLOCKS": "8", "MM_INGEST_TARGET_BASES": "200000"}):
        r = subprocess.run([BIN, "view", "--gpu-ingest"] + _args(kw) + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert b"[gpu-ingest]" in r.stderr
        got = r.stdout.decode()
        assert got == r0.stdout.decode(), (exp, env)
        assert got == want if exact else sorted(got.splitlines()) == sorted(want.splitlines())
        pick = lambda err: [l.split("] ", 1)[1] for l in err.decode().splitlines() if "] total " in l]
        assert pick(r.stderr) == pick(r0.stderr)


@pytest.mark.parametrize("bam,ctg,c", [("example-ont.bam", "chr22", "*"), ("eb.bam", "chr1", "*"), ("dRNA.bam", "chr22", "*[*]"), ("example-hifi.bam", "chr22", "*[CG]")])
def test_cli_wildcard_freq_with_the_device_reader(bam, ctg, c, fastas):
    """`minimod freq -c '*' --gpu-ingest`: the code table is filled from the census of every device batch in the order the host's walk
    fills it (code indices, and with them the replayed tie order, are the same): the bytes of the host threads' run, of the host's serial
    replay, and -- with groups of 8 blocks -- of several batches whose codes arrive one batch after the other"""
    tail = [fastas[ctg], os.path.join(GOLDEN, "data", bam)]
    base = [BIN, "freq", "-c", c]
    r0 = subprocess.run(base + ["--no-gpu-ingest", "--host-replay"] + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r0.returncode == 0 and len(r0.stdout) > 1000, r0.stderr.decode()[-2000:]
    for extra, env in ((["--gpu-ingest"], {}), (["--gpu-ingest"], {"MM_INGEST_MAX_BLOCKS": "8", "MM_INGEST_TARGET_BASES": "200000"}), (["--gpu-ingest", "--insertions"], {})):
        r = subprocess.run(base + extra + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert b"[gpu-ingest]" in r.stderr and b"on the device" in r.stderr
        if "--insertions" in extra:
            r1 = subprocess.run(base + ["--no-gpu-ingest", "--host-replay", "--insertions"] + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            assert r1.returncode == 0 and r.stdout == r1.stdout
        else:
            assert r.stdout == r0.stdout, (bam, c, env)


def test_cli_view_names_a_failing_read_with_the_device_reader(tmp_path):
    """view's error message with the device-side reader: the failing read by its index in the reference's -K batch, as with the host reader"""
    from minimod_amd import synth
    from oracle import pybam
    from tests.cases import KAT_REF, KAT_SEQ
    recs = [pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+m?,0,0;", [255, 3]) for _ in range(11)]
    recs[9] = pybam.make_record(0, 2, 0, KAT_SEQ, "3H20M", "C+m?,0,0;", [255, 3])
    bam, fa = str(tmp_path / "h.bam"), str(tmp_path / "h.fa")
    synth.write_bam(bam, [("chrT", len(KAT_REF))], [pybam.flatten(recs)], filter_fodder=False)
    synth.write_fasta(fa, "chrT", np.frombuffer(KAT_REF.encode(), dtype=np.uint8))
    for extra, want in ((["-K", "4"], b"read 1 of the batch"), (["-K", "7"], b"read 2 of the batch")):
        for ing in ("--gpu-ingest", "--no-gpu-ingest"):
            r = subprocess.run([BIN, "view", ing] + extra + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            assert r.returncode == 1 and b"Hard clipping found in " + want + b" (contig chrT, pos 2)" in r.stderr, (ing, extra, r.stderr.decode()[-1500:])


def test_cli_takes_more_than_thirteen_entries(fastas, chr22):
    """ABI 6: 32 -c entries over at most 13 different contexts (the reference has no limit, src/minimod.h:114; rounds 1 - 4 stopped at 13).
    Eighteen entries over three contexts on the reference's ONT example: the rows of the oracle, and the reference's tie order from the
    device-side replay byte for byte what the host's serial restatement prints; a 33rd entry is refused with a message; sixteen DIFFERENT contexts are counted (round 6)"""
    from oracle import oracle as O
    bam = os.path.join(GOLDEN, "data", "example-ont.bam")
    codes = ["m", "h", "a", "21839", "76792", "b", "c", "d", "e", "f", "g", "i", "j", "k", "l", "n", "o", "p"]
    ctxs = ["CG", "A", "*"]
    c = ",".join("%s[%s]" % (x, ctxs[i % 3]) for i, x in enumerate(codes))
    m = ",".join(["0.8", "0.7", "0.5"][i % 3] for i in range(len(codes)))
    base = [BIN, "freq", "-c", c, "-m", m]
    tail = [fastas["chr22"], bam]
    r = subprocess.run(base + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"on the device" in r.stderr
    rh = subprocess.run(base + ["--host-replay"] + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert rh.returncode == 0 and rh.stdout == r.stdout
    rows, names, wcodes = O.freq(bam, chr22, c=c, m=m)
    want = O.format_rows(rows, names, wcodes)
    assert len(want) > 50000 and sorted(r.stdout.decode().splitlines()) == sorted(want.splitlines())
    rb = subprocess.run(base + ["-b", "--gpu-ingest"] + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)   # (bedmethyl: the m rows only)
    assert rb.returncode == 0 and sorted(rb.stdout.decode().splitlines()) == sorted(O.format_rows(rows, names, wcodes, bedmethyl=True).splitlines())
    too_many = ",".join("%d[CG]" % (1000 + i) for i in range(33))
    r33 = subprocess.run([BIN, "freq", "-c", too_many] + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r33.returncode == 1 and b"At most 32 modification codes" in r33.stderr
    # round 6: a 14th, a 16th different context is counted like the others (site indices built in passes of thirteen contexts)
    ctx16 = ["CG", "A", "C", "CT", "CC", "T", "G", "AC", "GC", "TA", "CA", "GG", "TT", "AG", "*", "CGA"]
    c16 = ",".join("%s[%s]" % (x, ctx16[i]) for i, x in enumerate(codes[:16]))
    r16 = subprocess.run([BIN, "freq", "--canonical-order", "-c", c16] + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r16.returncode == 0, r16.stderr.decode()[-2000:]
    rows16, names16, wcodes16 = O.freq(bam, chr22, c=c16)
    assert sorted(r16.stdout.decode().splitlines()) == sorted(O.format_rows(rows16, names16, wcodes16).splitlines()) and len(r16.stdout) > 50000


# ---- many processes on one GPU (round 6: csrc/devmem.h, profiles/r6_site_index_root_cause.txt)
def test_twelve_clis_at_once(tmp_path):
    """Twelve CLIs at once on one GPU, each on an input of its own, ten runs each: every run exits 0 with the bytes of a quiet run on the same
    input.  (Rounds 4 and 5 printed a wrong strand's rows once in a hundred such runs: the site index's block counts were read through the
    translation of the buffer mm_freq_create had just freed.  tools/cli_stress.py is the same loop for longer campaigns.)"""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    from minimod_amd import synth
    from oracle import oracle as O
    W, K = 12, 10
    cmds, want = [], []
    for w in range(W):
        ref = synth.reference(13 + w, (2 << 20) - w * 50000)
        bs = [synth.batch(ref, i * 200, 200, seed=3 + 11 * w, n_reads_total=600) for i in range(3)]
        bam, fa = str(tmp_path / ("s%d.bam" % w)), str(tmp_path / ("s%d.fa" % w))
        synth.write_bam(bam, [("chrS", len(ref))], bs)
        synth.write_fasta(fa, "chrS", ref)
        cmds.append([BIN, "freq", "-b", "-c", "m[CG]", "-m", "0.8", "-K", "512", "-B", "100M", "-t", "4", fa, bam])
        quiet = subprocess.run(cmds[-1], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert quiet.returncode == 0, quiet.stderr.decode()[-2000:]
        if w == 0:   # (one of the inputs against the oracle: the quiet runs are the reference's bytes)
            orc = O.Oracle([("m", "CG")], [0.8], ["chrS"])
            orc.add_contig("chrS", ref)
            for b in bs:
                orc.process(b, threads=4)
            assert quiet.stdout.decode() == O.format_rows(orc.rows(), ["chrS"], orc.code_names(), bedmethyl=True)
        want.append(hashlib.md5(quiet.stdout).hexdigest())

    def worker(w):
        bad = []
        for k in range(K):
            r = subprocess.run(cmds[w], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            if r.returncode != 0 or hashlib.md5(r.stdout).hexdigest() != want[w] or b"failed its check" in r.stderr:
                bad.append((w, k, r.returncode, r.stderr.decode(errors="replace")[-800:]))
        return bad
    with ThreadPoolExecutor(max_workers=W) as ex:
        bad = [b for bl in ex.map(worker, range(W)) for b in bl]
    assert not bad, "%d of %d runs went wrong: %r" % (len(bad), W * K, bad[:2])


def test_a_site_index_that_fails_its_check_is_rebuilt_once_and_then_refused(tmp_path):
    """The guard behind the site index (k_site_check, mm_freq_create): an index whose ranks do not add up is never counted into.  MM_SITE_FAULT=1
    damages the first build (the run repairs it and prints the right bytes), =2 both (the run refuses)."""
    from minimod_amd import synth
    ref = synth.reference(5, 1 << 20)
    bs = [synth.batch(ref, 0, 200, seed=9, n_reads_total=200)]
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    synth.write_bam(bam, [("chrS", len(ref))], bs)
    synth.write_fasta(fa, "chrS", ref)
    cmd = [BIN, "freq", "-b", "-c", "m[CG]", "-m", "0.8", fa, bam]
    good = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert good.returncode == 0 and len(good.stdout) > 10000 and b"failed its check" not in good.stderr
    once = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, MM_SITE_FAULT="1"))
    assert once.returncode == 0 and once.stdout == good.stdout and b"failed its check" in once.stderr and b"building it once more" in once.stderr
    twice = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, MM_SITE_FAULT="2"))
    assert twice.returncode != 0 and twice.stdout == b"" and b"refused" in twice.stderr


# ---- where the device-side replay of the row order gives up, the run goes to the host's replay: the reference's bytes (src/mod.c:59-93,655-663)
def _tied_inputs(tmp_path, records, ref):
    from minimod_amd import synth
    from oracle import pybam
    bam, fa = str(tmp_path / "t.bam"), str(tmp_path / "t.fa")
    synth.write_bam(bam, [("chrT", len(ref))], [pybam.flatten(records)], filter_fodder=False)
    synth.write_fasta(fa, "chrT", np.frombuffer(ref, dtype=np.uint8))
    return bam, fa


def _fallback_equals_host_replay(cmd, fa, bam):
    host = subprocess.run(cmd + ["--host-replay", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert host.returncode == 0 and b"on the host" in host.stderr, host.stderr.decode()[-2000:]
    dflt = subprocess.run(cmd + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert dflt.returncode == 0, dflt.stderr.decode()[-2000:]
    assert b"the run starts again with --host-replay" in dflt.stderr, dflt.stderr.decode()[-2000:]
    assert len(host.stdout) > 1000 and dflt.stdout == host.stdout
    canon = subprocess.run(cmd + ["--canonical-order", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert canon.returncode == 0 and sorted(canon.stdout.splitlines()) == sorted(host.stdout.splitlines())
    return host.stdout, canon.stdout


def test_cli_haplotype_tag_above_61_goes_to_the_host_replay(tmp_path):
    """HP:C:62 -- a haplotype the device replay's 6-bit field does not hold: the run is handed to --host-replay (tieorder.c), not printed in the
    canonical order with a warning."""
    from oracle import pybam
    rng = np.random.default_rng(7)
    ref = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=6000).tobytes())
    recs = []
    for i in range(60):
        pos = int(rng.integers(0, 3000))
        seq = ref[pos:pos + 2000].decode()
        cs = [j for j, ch in enumerate(seq) if ch == "C"]
        skips = [cs[0]] and [0] * min(len(cs), 150)
        mm = "C+m?" + "".join(",%d" % k for k in skips) + ";C+h?" + "".join(",%d" % k for k in skips) + ";"
        ml = [int(x) for x in rng.integers(0, 256, size=2 * len(skips))]
        recs.append(pybam.make_record(0, pos, 0, seq, "2000M", mm, ml, hp=[1, 2, 62, 63][i % 4], qname=b"r%d" % i))
    recs.sort(key=lambda r: r.pos)
    bam, fa = _tied_inputs(tmp_path, recs, ref)
    host, canon = _fallback_equals_host_replay([BIN, "freq", "--haplotypes", "-c", "m[*],h[*]", "-m", "0.5,0.5", "-K", "16"], fa, bam)
    assert host != canon   # (the two orders differ on this input: the test means something)


def test_cli_read_with_300k_calls_goes_to_the_host_replay(tmp_path):
    """A read with more than 2^18 calls (the device replay numbers a read's calls with 18 bits): handed to --host-replay."""
    from oracle import pybam
    n = 270000   # (two groups: of a multi-code group "C+mh" only the suffix "h" is a code the reference looks up, src/mod.c:1151)
    ref = b"C" * (n + 64) + b"ACGT" * 16
    mm = "C+m?" + ",0" * n + ";C+h?,5,7;"
    rng = np.random.default_rng(11)
    ml = [int(x) for x in rng.integers(0, 256, size=n + 2)]
    recs = [pybam.make_record(0, 0, 0, "C" * n, "%dM" % n, mm, ml, qname=b"long"),
            pybam.make_record(0, 10, 16, "G" * 50 + "ACGT", "54M", "C+mh?,0;", [200, 10], qname=b"short")]
    bam, fa = _tied_inputs(tmp_path, recs, ref)
    _fallback_equals_host_replay([BIN, "freq", "-c", "m[*],h[*]", "-m", "0.5,0.5"], fa, bam)


def test_cli_refuses_what_neither_replay_takes_unless_the_canonical_order_is_asked_for(fastas):
    """-K 2097152 and more with rows that can tie: neither replay numbers such a batch's reads -- the run refuses (exit 1) instead of printing another
    order than the reference's; --canonical-order prints."""
    bam = os.path.join(GOLDEN, "data", "example-ont.bam")
    base = [BIN, "freq", "-c", "m[CG],h[CG]", "-K", "2097152"]
    r = subprocess.run(base + [fastas["chr22"], bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 1 and r.stdout == b"" and b"--canonical-order" in r.stderr
    c = subprocess.run(base + ["--canonical-order", fastas["chr22"], bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert c.returncode == 0 and len(c.stdout) > 1000
