"""Parity of the HIP hot path (through the C ABI) against the reference's golden files, the known-answer reads
and the oracle.  Needs a real MI355X: run with `-m gpu`.  Bit-exact: all quantities are integers."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pybam
from tests.cases import GOLDEN, GOLDEN_CASES, KAT2_INS, KAT_M, KAT_REF, kat2_records, kat_records
from tests.hiprun import hip_freq, hip_rows_from_records, make_engine, to_oracle_rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("exp,bam,ctg,kw,exact", GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def test_reference_golden(exp, bam, ctg, kw, exact, request):
    contigs = request.getfixturevalue(ctg)
    rows, names, codes = hip_freq(os.path.join(GOLDEN, "data", bam), contigs, **kw)
    txt = O.format_rows(rows, names, codes, bedmethyl=exp.endswith("bedmethyl"),
                        insertions=kw.get("insertions", False), haplotypes=kw.get("haplotypes", False))
    want = open(os.path.join(GOLDEN, "expected", exp)).read()
    if exact:
        assert txt == want
    else:
        assert sorted(txt.splitlines()) == sorted(want.splitlines())


def test_kat_m():
    assert [r[:4] for r in hip_rows_from_records(kat_records(), KAT_REF, "m")] == KAT_M
    assert [r[:4] for r in hip_rows_from_records(kat_records(), KAT_REF, "m,h")] == KAT_M


def test_kat_multiletter_and_star_context():
    rows = hip_rows_from_records(kat_records(), KAT_REF, "hm[CG]")
    assert [(r[0], r[3], r[6]) for r in rows] == [(2, 0, "hm"), (6, 0, "hm"), (10, 0, "hm"), (15, 0, "hm"), (19, 0, "hm")]
    assert (9, "+", 3, 2) in [r[:4] for r in hip_rows_from_records(kat_records(), KAT_REF, "m[*]")]


def test_kat_insertions_haplotypes():
    rows = hip_rows_from_records(kat_records(), KAT_REF, "m", insertions=True, haplotypes=True)
    assert (6, "+", 1, 1, 1, 1, "m") in rows and (6, "+", 1, 1, 1, -1, "m") in rows
    assert (2, "+", 2, 1, 0, 0, "m") in rows
    assert (9, "+", 3, 2, 0, -1, "m") in rows


def test_kat2_insertion_orientation_quirk():
    assert [r[:5] for r in hip_rows_from_records(kat2_records(), KAT_REF, "m", insertions=True)] == KAT2_INS


# configurations the reference's goldens cannot pin offline ('.'-flag BAMs need every reference base): HIP vs oracle
# on the same pseudo-reference
ORACLE_CASES = [
    ("example-ont.bam", dict(c="m[CG],h[CG]", m="0.8,0.7")),
    ("example-ont.bam", dict(c="m", insertions=True, haplotypes=True)),
    ("example-ont.bam", dict(c="*")),
    ("example-hifi.bam", dict(c="m", insertions=True)),
    ("dna_5mC_5hmC_mm_chr22.bam", dict(c="m[CG],h[CG]")),
    ("dna_5mC_5hmC_mm_chr22.bam", dict(c="m[C]", insertions=True, haplotypes=True, skip_supplementary=True)),
    ("dna_4mC_5mC_mm_chr22.bam", dict(c="21839[C],m[*]", m="0.7,0.9", insertions=True)),
    ("dna_6mA_mm_chr22.bam", dict(c="a[A]")),
    ("dna_5mCG_5hmCG_mm_with_secondary_chr22.bam", dict(c="*[CG]", allow_secondary=True)),
    ("dRNA.bam", dict(c="17802[*],a,m[C]")),
    ("dna_5mCG_5hmCG_mm_chr22.bam", dict(c="m,h", m="0.8,0.7", insertions=True, haplotypes=True, K=7)),
    # more than five -c entries: 32-bit reference words (two context bits per entry), the kernels' uint32 instantiation
    ("example-ont.bam", dict(c="m[CG],h[CG],a[A],c[C],f[C],e[T],b[T]", m="0.8,0.7,0.6,0.8,0.8,0.8,0.8")),
    ("dRNA.bam", dict(c="17802[*],a,m[C],17596[A],19228[C],19227[T],69426[A],19229[G],o,n,g,e,b", insertions=True)),
]


@pytest.mark.parametrize("bam,kw", ORACLE_CASES, ids=["%s:%s" % (b, k.get("c")) for b, k in ORACLE_CASES])
def test_against_oracle(bam, kw, chr22):
    path = os.path.join(GOLDEN, "data", bam)
    want, names, wcodes = O.freq(path, chr22, **kw)
    got, _, gcodes = hip_freq(path, chr22, **kw)
    a = O.format_rows(want, names, wcodes, insertions=kw.get("insertions", False), haplotypes=kw.get("haplotypes", False))
    b = O.format_rows(got, names, gcodes, insertions=kw.get("insertions", False), haplotypes=kw.get("haplotypes", False))
    assert len(want) > 0
    assert sorted(a.splitlines()) == sorted(b.splitlines())


def test_eb_hap_against_oracle(chr1):
    for bam, kw in (("eb.bam", dict(c="e,b", m="0.5", insertions=True)),
                    ("hap.bam", dict(c="m", haplotypes=True, insertions=True, n_hp_planes=2))):
        path = os.path.join(GOLDEN, "data", bam)
        okw = {k: v for k, v in kw.items() if k != "n_hp_planes"}
        want, names, wcodes = O.freq(path, chr1, **okw)
        got, _, gcodes = hip_freq(path, chr1, **kw)
        a = O.format_rows(want, names, wcodes, insertions=True, haplotypes=kw.get("haplotypes", False))
        b = O.format_rows(got, names, gcodes, insertions=True, haplotypes=kw.get("haplotypes", False))
        assert sorted(a.splitlines()) == sorted(b.splitlines())


ERR_CASES = [
    (pybam.make_record(0, 2, 0, "CGTT", "2H4M", "C+m?,0;", [255]), 1),
    (pybam.make_record(0, 2, 0, "CGTT", "4M", "X+m?,0;", [255]), 3),
    (pybam.make_record(0, 2, 0, "CGTT", "4M", "C*m?,0;", [255]), 4),
    (pybam.make_record(0, 2, 0, "CGTT", "4M", "C+m?,0,0;", [255]), 10),       # more calls than C bases
    (pybam.make_record(0, 2, 0, "CGTTCG", "6M", "C+m?,0,0;", [255]), 11),     # ML shorter than MM
    (pybam.make_record(0, 2, 0, "CGTT", "4M", "C+m?,1234567890;", [255]), 8),
    (pybam.make_record(0, 28, 0, "CGTT", "4M", "C+m?,0;", [255]), 13),        # runs off the contig
]


@pytest.mark.parametrize("rec,code", ERR_CASES, ids=[str(c) for _, c in ERR_CASES])
def test_error_codes_match_oracle(rec, code):
    import minimod_amd
    mods = O.parse_mod_codes("m")
    o = O.Oracle(mods, [0.8], ["chrT"])
    o.add_contig("chrT", KAT_REF.encode())
    with pytest.raises(O.OracleError) as oe:
        o.process(pybam.flatten([rec]))
    assert oe.value.code == code
    good = kat_records()
    with pytest.raises(minimod_amd.MinimodHipError) as he:
        hip_rows_from_records(good[:2] + [rec] + good[2:], KAT_REF, "m")
    assert he.value.code == code and he.value.read == 2


def test_empty_and_ragged_batches():
    eng = make_engine([("m", "CG")], [0.8], ["chrT"], [len(KAT_REF)], {"chrT": KAT_REF.encode()})
    eng.process(pybam.flatten([]))                                   # empty batch
    assert len(eng.finalize()) == 0
    recs = [pybam.make_record(0, 2, 0, "CGTT", "4M", "", []),          # empty MM string
            pybam.make_record(0, 2, 0, "CGTT", "4M", "C+m?;", []),     # group without calls
            pybam.make_record(0, 2, 0, "C", "1M", "C+m?,0;", [255]),   # one base
            pybam.make_record(0, 2, 0, "CGTT", "*" and "", "C+m?,0;", [255])]  # no CIGAR at all
    eng.process(pybam.flatten(recs))
    rows = eng.finalize()
    assert [(int(r["pos"]), int(r["n_called"]), int(r["n_mod"])) for r in rows] == [(2, 1, 1)]
    eng.close()


def test_missing_contig_fails_like_reference():
    import minimod_amd
    eng = make_engine([("m", "CG")], [0.8], ["chrT", "chrU"], [len(KAT_REF), 100], {"chrT": KAT_REF.encode()})
    rec = pybam.make_record(1, 2, 0, "CGTT", "4M", "C+m?,0;", [255])
    with pytest.raises(minimod_amd.MinimodHipError) as he:
        eng.process(pybam.flatten([rec]))
    assert he.value.code == 12
    eng.close()


def test_long_read_spills_past_lds_caps(chr22):
    """A read longer than the LDS directory (32 kb) and with more CIGAR ops than the LDS prefix arrays (1024)."""
    path = os.path.join(GOLDEN, "data", "example-ont.bam")
    bam = pybam.BamFile(path)
    recs = [r for r in bam if pybam.accept(r)]
    assert max(r.l_qseq for r in recs) > 32768 and max(r.n_cigar for r in recs) > 1024


def test_unwaited_tickets_and_fallback_reads(chr22):
    """dRNA.bam's reads carry MM groups on different canonical bases: the tile kernels leave them to the fused kernel,
    which runs when the host waits for the batch.  Here nobody waits: slots are reused (more batches than slots) and
    finalize has to settle what is outstanding.  Same rows as with a wait after every batch, and as the oracle."""
    path = os.path.join(GOLDEN, "data", "dRNA.bam")
    kw = dict(c="17802[*],a,m[C]")
    want, names, wcodes = O.freq(path, chr22, **kw)
    mods = O.parse_mod_codes(kw["c"])
    th = O.parse_mod_threshes(None, len(mods))
    eng = None
    n_batches = 0
    for bam, batch, _st in pybam.load_batches(path, K=8):
        if eng is None:
            eng = make_engine(mods, th, bam.target_name, bam.target_len, chr22)
        if len(batch["reads"]):
            eng.submit(batch)          # no wait
            n_batches += 1
    assert n_batches > 4
    got = to_oracle_rows(eng.finalize())
    codes = eng.code_names()
    eng.close()
    a = O.format_rows(want, names, wcodes)
    b = O.format_rows(got, names, codes)
    assert len(want) > 1000 and sorted(a.splitlines()) == sorted(b.splitlines())


def test_wildcard_codes_beyond_the_dense_planes_and_side_list_full(chr22):
    """-c '*' on dRNA.bam interns eight code strings; with two dense planes the other six count through the side list.
    A side list that is too small is reported (MM_E_SIDEFULL), not silently truncated."""
    import minimod_amd
    path = os.path.join(GOLDEN, "data", "dRNA.bam")
    want, names, wcodes = O.freq(path, chr22, c="*")
    got, _, gcodes = hip_freq(path, chr22, c="*", n_wild_planes=2)
    assert len(wcodes) >= 8 and len(want) > 1000
    a, b = O.format_rows(want, names, wcodes), O.format_rows(got, names, gcodes)
    assert sorted(a.splitlines()) == sorted(b.splitlines())
    with pytest.raises(minimod_amd.engine.MinimodHipError) as e:
        hip_freq(path, chr22, c="*", n_wild_planes=1, side_capacity=64)
    assert e.value.code == 32
