"""Parity of the HIP `view` path (through the C ABI) against the reference's view goldens, known-answer reads and the
oracle's view mode.  Needs a real MI355X: run with `-m gpu`.  Bit-exact: all quantities are integers."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pybam
from tests.cases import GOLDEN, KAT_REF, KAT_SEQ, VIEW_CASES, kat2_records, kat_records
from tests.hiprun import hip_view, hip_view_from_records

pytestmark = pytest.mark.gpu


def _text(res, kw):
    rows, qnames, names, codes = res
    return O.format_view(rows, qnames, names, codes, insertions=kw.get("insertions", False),
                         haplotypes=kw.get("haplotypes", False))


@pytest.mark.parametrize("exp,bam,ctg,kw,exact", VIEW_CASES, ids=["view-" + c[0] for c in VIEW_CASES])
def test_reference_view_golden(exp, bam, ctg, kw, exact, request):
    contigs = request.getfixturevalue(ctg)
    txt = _text(hip_view(os.path.join(GOLDEN, "data", bam), contigs, **kw), kw)
    want = open(os.path.join(GOLDEN, "expected", exp)).read()
    if exact:
        assert txt == want
    else:
        assert sorted(txt.splitlines()) == sorted(want.splitlines())


def test_kat_view_rows():
    got = hip_view_from_records(kat_records(), KAT_REF, "m")
    # read 0: C+hm -> 'm' is the second letter: ML bytes 250,5,250,5,250,5 at CG sites only (context CG)
    assert [g for g in got if g[0] == 0] == [(0, 2, 0, "m", 250, 0), (0, 6, 4, "m", 5, 0), (0, 10, 8, "m", 5, 0),
                                             (0, 15, 13, "m", 250, 0), (0, 19, 17, "m", 5, 0)]
    # read 1: '.' group, second C listed: implicit calls carry 0
    assert [g for g in got if g[0] == 1] == [(1, 2, 0, "m", 0, 0), (1, 6, 4, "m", 255, 0), (1, 10, 8, "m", 0, 0),
                                             (1, 15, 13, "m", 0, 0), (1, 19, 17, "m", 0, 0)]


def test_view_first_entry_wins():
    recs = [pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+m.,1;C+m?,1;", [255, 7], qname=b"dup")]
    got = hip_view_from_records(recs, KAT_REF, "m")
    assert [(g[1], g[2], g[4]) for g in got] == [(2, 0, 0), (6, 4, 255), (10, 8, 0), (15, 13, 0), (19, 17, 0)]
    recs = [pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+m?,1;C+m.,1;", [7, 255], qname=b"dup")]
    got = hip_view_from_records(recs, KAT_REF, "m")
    assert [(g[1], g[2], g[4]) for g in got] == [(2, 0, 0), (6, 4, 7), (10, 8, 0), (15, 13, 0), (19, 17, 0)]


def _oracle_view_records(recs, c, **kw):
    mods = O.parse_mod_codes(c)
    o = O.Oracle(mods, O.parse_mod_threshes(None, len(mods)), ["chrT"], **kw)
    o.set_view(True)
    o.add_contig("chrT", KAT_REF.encode())
    o.process(pybam.flatten(recs))
    codes = o.code_names()
    return [(int(r["read"]), int(r["pos"]), int(r["read_pos"]), codes[r["code"]], int(r["prob"]), int(r["ins_off"]))
            for r in o.view_rows()]


@pytest.mark.parametrize("c,kw", [("m", dict()), ("m,h", dict()), ("hm[*]", dict()), ("m", dict(insertions=True)),
                                  ("m[*]", dict(insertions=True, haplotypes=True))])
def test_kat_view_against_oracle(c, kw):
    recs = kat_records() + kat2_records()
    assert hip_view_from_records(recs, KAT_REF, c, **kw) == _oracle_view_records(recs, c, **kw)


ORACLE_VIEW_CASES = [
    ("example-ont.bam", dict(c="m[CG],h[CG]")),
    ("example-ont.bam", dict(c="m", insertions=True, haplotypes=True)),
    ("example-hifi.bam", dict(c="m", insertions=True)),
    ("dna_5mC_5hmC_mm_chr22.bam", dict(c="m[C]", insertions=True, haplotypes=True, skip_supplementary=True)),
    ("dna_4mC_5mC_mm_chr22.bam", dict(c="21839[C],m[*]", insertions=True)),
    ("dna_6mA_mm_chr22.bam", dict(c="a[A]")),
    ("dna_5mCG_5hmCG_mm_with_secondary_chr22.bam", dict(c="*[CG]", allow_secondary=True)),
    ("dRNA.bam", dict(c="17802[*],a,m[C]")),
    ("dna_5mCG_5hmCG_mm_chr22.bam", dict(c="m,h", insertions=True, haplotypes=True, K=7)),
    ("example-ont.bam", dict(c="m[CG],h[CG],a[A],c[C],f[C],e[T],b[T]")),      # > 5 entries: 32-bit reference words
    ("dRNA.bam", dict(c="17802[*],a,m[C],17596[A],19228[C],19227[T],69426[A],19229[G],o,n,g,e,b")),
]


@pytest.mark.parametrize("bam,kw", ORACLE_VIEW_CASES, ids=["%s:%s" % (b, k.get("c")) for b, k in ORACLE_VIEW_CASES])
def test_view_against_oracle(bam, kw, chr22):
    path = os.path.join(GOLDEN, "data", bam)
    want = O.view(path, chr22, **kw)
    got = hip_view(path, chr22, **kw)
    assert len(want[0]) > 0
    a, b = _text(want, kw), _text(got, kw)
    if "*" in kw["c"].split("[")[0]:          # wildcard: code indices are interning order, ties may swap
        assert sorted(a.splitlines()) == sorted(b.splitlines())
    else:
        assert a == b


def test_view_region_overflow_reruns(chr22):
    """A record buffer too small for the batch: fetch grows it and runs the batch again; same rows."""
    path = os.path.join(GOLDEN, "data", "example-ont.bam")
    kw = dict(c="m,h")
    want = _text(hip_view(path, chr22, **kw), kw)
    got = _text(hip_view(path, chr22, view_cap=16, **kw), kw)
    assert got == want


GATHER_VIEW_WORKER = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np
import torch
torch.zeros(1, device="cuda")
import minimod_amd
from minimod_amd import synth
from oracle import oracle as O
ref = synth.reference(21, 4 << 20)
bs = [synth.batch(ref, i * 300, 300, seed=9, n_reads_total=1500, with_order=False, dot_fraction=0.3) for i in range(5)]
whole = synth.concat(bs)
dev = {k: torch.from_numpy(whole[k].view(np.uint8).reshape(-1)).cuda() for k in ("reads", "cigar", "seq", "mm", "ml")}
def window(i):
    return dict(reads=dev["reads"].data_ptr() + 64 * 300 * i, cigar=dev["cigar"].data_ptr(), seq=dev["seq"].data_ptr(), mm=dev["mm"].data_ptr(),
                ml=dev["ml"].data_ptr(), n_reads=300, n_cigar_words=len(whole["cigar"]), n_seq_bytes=len(whole["seq"]), n_mm_bytes=len(whole["mm"]),
                n_ml_bytes=len(whole["ml"]), max_n_cigar=int(bs[i]["max_n_cigar"]), max_l_qseq=int(bs[i]["max_l_qseq"]))
mods, th = [("m", "CG"), ("h", "CG")], [0.8, 0.7]
# the oracle on the windows of a group taken as one batch: `read` counts from the group's first read
def oracle_rows(lo, hi):
    orc = O.Oracle(mods, th, ["chrS"]); orc.set_view(True); orc.add_contig("chrS", ref)
    orc.process(synth.concat(bs[lo:hi]), threads=8)
    return orc.view_rows()
vk = lambda r, io: list(zip(r["read"].tolist(), r["pos"].tolist(), r["read_pos"].tolist(), r["code"].tolist(), r[io].tolist(), r["prob"].tolist()))
eng = minimod_amd.FreqEngine([(c, x, t) for (c, x), t in zip(mods, th)], [("chrS", len(ref), ref)], view=True, coalesce=3, stream_mode=3)
tickets = [eng.submit_device(window(i)) for i in range(5)]
out = {"tickets": tickets, "sizes": []}
groups = [(0, 3), (3, 5)]
ok, rows = True, 0
for (lo, hi), tk in zip(groups, list(dict.fromkeys(tickets))):
    out["sizes"].append(eng.ticket_batches(tk))
    got = eng.fetch_view(tk)
    want = oracle_rows(lo, hi)
    rows += len(want)
    ok = ok and vk(got, "ins_offset") == vk(want, "ins_off")
eng.close()
out["equal"], out["rows"] = ok, rows
print(json.dumps(out))
'''


def test_gathered_windows_in_view_mode():
    """mm_freq_opts_t.coalesce in view mode: consecutive windows of a resident read set share a launch and a ticket; the
    ticket's rows are those of the group in print_view_output order, `read` counted from the group's first read -- the
    oracle's rows on the same windows taken as one batch, element for element (a third of the reads carry '.' groups)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", GATHER_VIEW_WORKER % root], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert res["tickets"][0] == res["tickets"][2] != res["tickets"][3] and res["sizes"] == [3, 2]
    assert res["rows"] > 10000 and res["equal"], res
