"""Pin the oracle (oracle/freq_oracle.c) against the reference's own golden files and the
hand-built known-answer reads.  CPU only."""
import os

import pytest

from oracle import oracle as O
from oracle import pybam
from tests.cases import GOLDEN, GOLDEN_CASES, VIEW_CASES, KAT2_INS, KAT_M, KAT_REF, KAT_SEQ, kat2_records, kat_records


@pytest.mark.parametrize("exp,bam,ctg,kw,exact", GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def test_reference_golden(exp, bam, ctg, kw, exact, request):
    contigs = request.getfixturevalue(ctg)
    rows, names, codes = O.freq(os.path.join(GOLDEN, "data", bam), contigs, **kw)
    txt = O.format_rows(rows, names, codes, bedmethyl=exp.endswith("bedmethyl"),
                        insertions=kw.get("insertions", False), haplotypes=kw.get("haplotypes", False))
    want = open(os.path.join(GOLDEN, "expected", exp)).read()
    if exact:
        assert txt == want
    else:
        assert sorted(txt.splitlines()) == sorted(want.splitlines())


@pytest.mark.parametrize("exp,bam,ctg,kw,exact", VIEW_CASES, ids=["view-" + c[0] for c in VIEW_CASES])
def test_reference_view_golden(exp, bam, ctg, kw, exact, request):
    contigs = request.getfixturevalue(ctg)
    rows, qnames, names, codes = O.view(os.path.join(GOLDEN, "data", bam), contigs, **kw)
    txt = O.format_view(rows, qnames, names, codes, insertions=kw.get("insertions", False),
                        haplotypes=kw.get("haplotypes", False))
    want = open(os.path.join(GOLDEN, "expected", exp)).read()
    if exact:
        assert txt == want
    else:
        assert sorted(txt.splitlines()) == sorted(want.splitlines())


def test_view_batch_invariance(chr22):
    bam = os.path.join(GOLDEN, "data", "example-ont.bam")
    a = O.view(bam, chr22, c="m,h", insertions=True, haplotypes=True, K=7)[0]
    b = O.view(bam, chr22, c="m,h", insertions=True, haplotypes=True, K=4096, threads=3)[0]
    assert (a == b).all()


def test_view_first_entry_wins_and_implicit_prob_zero():
    """add_view_entry keeps the first entry of a key (mod.c:931-946); '.'-mode implicit calls carry probability 0."""
    recs = [pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+m.,1;C+m?,1;", [255, 7], qname=b"dup")]
    mods = O.parse_mod_codes("m")
    o = O.Oracle(mods, O.parse_mod_threshes(None, 1), ["chrT"])
    o.set_view(True)
    o.add_contig("chrT", KAT_REF.encode())
    o.process(pybam.flatten(recs))
    got = [(int(r["pos"]), int(r["read_pos"]), int(r["prob"])) for r in o.view_rows()]
    # C at read 0,4,7,8,13,17 ; CG context at 0,4,8,13,17 ; listed C (rank 1) is read 4 -> prob 255, once
    assert got == [(2, 0, 0), (6, 4, 255), (10, 8, 0), (15, 13, 0), (19, 17, 0)]


def _run(recs, c, **kw):
    mods = O.parse_mod_codes(c)
    o = O.Oracle(mods, O.parse_mod_threshes(None, len(mods)), ["chrT"], **kw)
    o.add_contig("chrT", KAT_REF.encode())
    o.process(pybam.flatten(recs))
    codes = o.code_names()
    return [(int(r["pos"]), "+-"[r["strand"]], int(r["n_called"]), int(r["n_mod"]), int(r["ins_off"]), int(r["hp"]),
             codes[r["code"]]) for r in o.rows()]


def test_kat_m():
    assert [r[:4] for r in _run(kat_records(), "m")] == KAT_M
    assert [r[:4] for r in _run(kat_records(), "m,h")] == KAT_M  # 'h' of C+hm is looked up as "hm": never matches


def test_kat_multiletter_and_star_context():
    rows = _run(kat_records(), "hm[CG]")
    assert [(r[0], r[3], r[6]) for r in rows] == [(2, 0, "hm"), (6, 0, "hm"), (10, 0, "hm"), (15, 0, "hm"), (19, 0, "hm")]
    assert (9, "+", 3, 2) in [r[:4] for r in _run(kat_records(), "m[*]")]


def test_kat_insertions_haplotypes():
    rows = _run(kat_records(), "m", insertions=True, haplotypes=True)
    assert (6, "+", 1, 1, 1, 1, "m") in rows and (6, "+", 1, 1, 1, -1, "m") in rows   # inserted C, hp 1 and '*'
    assert (2, "+", 2, 1, 0, 0, "m") in rows                                          # untagged reads -> hp 0
    assert (9, "+", 3, 2, 0, -1, "m") in rows                                         # context ignored


def test_kat2_insertion_orientation_quirk():
    assert [r[:5] for r in _run(kat2_records(), "m", insertions=True)] == KAT2_INS


def test_batch_invariance(chr22):
    bam = os.path.join(GOLDEN, "data", "example-ont.bam")
    a = O.freq(bam, chr22, c="m,h", m="0.8,0.7", insertions=True, haplotypes=True, K=7)[0]
    b = O.freq(bam, chr22, c="m,h", m="0.8,0.7", insertions=True, haplotypes=True, K=4096, threads=3)[0]
    assert (a == b).all()


def test_hard_clip_is_an_error():
    rec = pybam.make_record(0, 2, 0, "CGTT", "2H4M", "C+m?,0;", [255])
    with pytest.raises(O.OracleError) as e:
        _run([rec], "m")
    assert e.value.code == 1
