"""Skip lists across tile boundaries: k_sum_tiles sums a list per character (digit weights, delimiter bitmaps) in
tiles of kTileChars characters (320; 256 in round 1) with a 16-character look-ahead; these cases move tokens of every width over every boundary
the kernel has (64-character sub-chunks, tile ends, the look-ahead, the end of the string) and place malformed tokens
there.  Zero-padded counts ("0007") are legal for the reference's parser (mod.c:1074-1081: digits folded one by one),
which is what lets short reads carry nine-character tokens.  HIP vs the oracle, bit-exact."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import pybam
from tests.hiprun import hip_rows_from_records

pytestmark = pytest.mark.gpu


def make_ref(rng, n):
    return "".join(rng.choice(list("ACGT"), size=n, p=[0.2, 0.3, 0.3, 0.2]))


def oracle_rows(recs, ref, c, **kw):
    mods = O.parse_mod_codes(c)
    o = O.Oracle(mods, O.parse_mod_threshes(None, len(mods)), ["chrT"], **kw)
    o.add_contig("chrT", ref.encode())
    o.process(pybam.flatten(recs))
    rows = o.rows()
    codes = o.code_names()
    o.close()
    return [(int(r["pos"]), "+-"[r["strand"]], int(r["n_called"]), int(r["n_mod"]), int(r["ins_off"]), int(r["hp"]),
             codes[r["code"]]) for r in rows]


def padded_list(rng, n_c, max_width, widths=None):
    """tokens with small values and random widths; returns (text without header, number of tokens)"""
    toks, used = [], 0
    while True:
        v = int(rng.integers(0, 3))
        if used + v + 1 > n_c:
            break
        used += v + 1
        w = int(rng.integers(1, max_width + 1)) if widths is None else widths[len(toks) % len(widths)]
        toks.append(str(v).zfill(w))
    return toks


def read_with_list(ref, pos, length, header, toks, rng):
    """one forward read over ref[pos:pos+length] with a single-code group `header` listing `toks`"""
    ml = [int(x) for x in rng.integers(0, 256, size=len(toks))]
    mm = header + "".join("," + t for t in toks) + ";"
    return pybam.make_record(0, pos, 0, ref[pos:pos + length], "%dM" % length, mm, ml)


@pytest.mark.parametrize("seed", range(12))
def test_random_widths_cross_every_boundary(seed):
    rng = np.random.default_rng(1000 + seed)
    ref = make_ref(rng, 6000)
    recs = []
    for i in range(24):
        length = int(rng.integers(1500, 5000))
        pos = int(rng.integers(0, len(ref) - length))
        seq = ref[pos:pos + length]
        header = ["C+m?", "C+m.", "C+m", "C+h?"][int(rng.integers(0, 4))]
        toks = padded_list(rng, seq.count("C"), int(rng.integers(1, 10)))
        recs.append(read_with_list(ref, pos, length, header, toks, rng))
    for c in ("m", "m,h", "m[*]"):
        assert hip_rows_from_records(recs, ref, c) == oracle_rows(recs, ref, c)


@pytest.mark.parametrize("width", range(1, 10))
def test_fixed_width_tokens_slide_over_the_tile_end(width):
    """every alignment of a width-w token against characters 255/256 and the 64-character sub-chunk ends"""
    rng = np.random.default_rng(50 + width)
    ref = make_ref(rng, 5000)
    recs = []
    for shift in range(0, width + 2):
        length = 4000
        seq = ref[:length]
        toks = padded_list(rng, seq.count("C"), width, widths=[width])
        toks[0] = toks[0].zfill(min(9, len(toks[0]) + shift)) if shift else toks[0]   # slides everything behind it
        recs.append(read_with_list(ref, 0, length, "C+m?", toks, rng))
    assert hip_rows_from_records(recs, ref, "m") == oracle_rows(recs, ref, "m")


def test_two_groups_and_unrequested_group_between():
    rng = np.random.default_rng(7)
    ref = make_ref(rng, 5000)
    recs = []
    for i in range(16):
        length = 3000
        seq = ref[i * 10:i * 10 + length]
        t1 = padded_list(rng, seq.count("C"), 6)
        t2 = padded_list(rng, seq.count("A"), 5)
        t3 = padded_list(rng, seq.count("C"), 4)
        mm = "C+m?" + "".join("," + t for t in t1) + ";A+a?" + "".join("," + t for t in t2) + ";C+h?" + "".join("," + t for t in t3) + ";"
        ml = [int(x) for x in rng.integers(0, 256, size=len(t1) + len(t2) + len(t3))]
        recs.append(pybam.make_record(0, i * 10, 0, seq, "%dM" % length, mm, ml))
    for c in ("m", "h", "m,h", "a[*]"):
        assert hip_rows_from_records(recs, ref, c) == oracle_rows(recs, ref, c)


def bad_list_cases():
    """(name, position of the malformed token in characters from the list start, token text)"""
    cases = []
    for at in (0, 60, 63, 64, 250, 254, 255, 256, 257, 300, 318, 319, 320, 321, 383, 384, 511, 512, 639, 640, 641):
        cases.append(("nondigit@%d" % at, at, "1x"))
        cases.append(("tenchars@%d" % at, at, "0000000001"))
        cases.append(("nondigit_in_long@%d" % at, at, "00x0000000001"))
        cases.append(("long_then_nondigit@%d" % at, at, "0000000000x"))
    return cases


@pytest.mark.parametrize("name,at,tok", bad_list_cases(), ids=[c[0] for c in bad_list_cases()])
def test_malformed_tokens_report_the_reference_error(name, at, tok):
    import minimod_amd
    rng = np.random.default_rng(3)
    ref = make_ref(rng, 5000)
    seq = ref[:4000]
    # a list of one-digit zero skips ("0," = 2 characters) up to `at`, then the bad token, then more good ones
    n_before = at // 2
    pad = at - 2 * n_before          # 0 or 1 extra character: widen the first token
    toks = ["0"] * n_before
    if pad and toks:
        toks[0] = "00"
    elif pad:
        tok = "0" + tok if tok[0].isdigit() else tok
    toks = toks + [tok] + ["0"] * 40
    mm = "C+m?" + "".join("," + t for t in toks) + ";"
    rec = pybam.make_record(0, 0, 0, seq, "4000M", mm, [200] * len(toks))
    mods = O.parse_mod_codes("m")
    o = O.Oracle(mods, [0.8], ["chrT"])
    o.add_contig("chrT", ref.encode())
    with pytest.raises(O.OracleError) as oe:
        o.process(pybam.flatten([rec]))
    o.close()
    assert oe.value.code in (8, 9)
    good = pybam.make_record(0, 0, 0, seq, "4000M", "C+m?,0,1;", [255, 0])
    with pytest.raises(minimod_amd.MinimodHipError) as he:
        hip_rows_from_records([good, rec, good], ref, "m")
    assert (he.value.code, he.value.read) == (oe.value.code, 1)


def test_list_without_trailing_semicolon_and_empty_tokens():
    rng = np.random.default_rng(11)
    ref = make_ref(rng, 3000)
    seq = ref[:2500]
    recs = []
    for n in (1, 63, 64, 127, 128, 129, 200):
        toks = ["0"] * n
        mm = "C+m?" + "".join("," + t for t in toks)           # no ';' at the end of the string
        recs.append(pybam.make_record(0, 0, 0, seq, "2500M", mm, [255] * n))
        mm2 = "C+m?,," + ",".join(toks) + ",,;"                  # empty tokens are skipped by the parser
        recs.append(pybam.make_record(0, 0, 0, seq, "2500M", mm2, [255] * n))
    assert hip_rows_from_records(recs, ref, "m") == oracle_rows(recs, ref, "m")


def test_tiles_whose_cigar_slice_spans_more_than_the_packed_range():
    """k_call_tiles packs a tile's CIGAR slice as 14-bit offsets from its first op; a long deletion / intron or a huge skip
    inside one tile exceeds that and the tile must fall back to the global arrays: same rows."""
    rng = np.random.default_rng(21)
    ref = make_ref(rng, 70000)
    recs = []
    # (a) 20 kb deletion and (b) 20 kb intron in the middle of densely listed calls
    for op in ("D", "N"):
        seq = ref[100:3100] + ref[23100:26100]
        n_c = seq.count("C")
        toks = ["0"] * n_c
        ml = [int(x) for x in rng.integers(0, 256, size=n_c)]
        recs.append(pybam.make_record(0, 100, 0, seq, "3000M20000%s3000M" % op, "C+m?" + "".join("," + t for t in toks) + ";", ml))
    # (c) one tile whose calls are 25 kb of read apart: a few listed calls, a skip over ~6000 Cs, a few more
    seq = ref[30000:65000]
    n_c = seq.count("C")
    toks = ["0"] * 20 + [str(n_c - 60)] + ["0"] * 20
    ml = [int(x) for x in rng.integers(0, 256, size=len(toks))]
    recs.append(pybam.make_record(0, 30000, 0, seq, "%dM" % len(seq), "C+m?" + "".join("," + t for t in toks) + ";", ml))
    # (d) the same with an insertion of 18 kb between the two clusters (read positions far apart, reference close)
    seq = ref[1000:1500] + make_ref(rng, 18000) + ref[1500:2000]
    n_c = seq.count("C")
    n_first = ref[1000:1500].count("C")
    toks = ["0"] * n_first + [str(n_c - n_first - ref[1500:2000].count("C"))] + ["0"] * (ref[1500:2000].count("C") - 1)
    ml = [int(x) for x in rng.integers(0, 256, size=len(toks))]
    recs.append(pybam.make_record(0, 1000, 0, seq, "500M18000I500M", "C+m?" + "".join("," + t for t in toks) + ";", ml))
    for c, kw in (("m", {}), ("m[*]", {}), ("m", dict(insertions=True))):
        assert hip_rows_from_records(recs, ref, c, **kw) == oracle_rows(recs, ref, c, **kw)


@pytest.mark.parametrize("n_listed", [90, 159, 160, 161, 320, 321, 700])
def test_unrequested_group_listing_more_bases_than_the_read_has(n_listed):
    """The reference asserts the read position of every listed base, also in groups -c did not ask for (mod.c:1116): a
    list that runs past the read's last base of that kind fails the read.  Lists of "0," tokens end on, before and behind
    tile boundaries (a 320-character tile holds 160 of them): only the last tile of an unrequested list does the check."""
    import minimod_amd
    rng = np.random.default_rng(5)
    ref = make_ref(rng, 4000)
    seq = ref[:3000]
    n_a = seq.count("A")
    good = pybam.make_record(0, 0, 0, seq, "3000M", "C+m?,0,1;", [255, 0])
    # fine: the A group lists the read's first n_listed As;  bad: its last token skips past the read's last A
    ok_toks = ["0"] * min(n_listed, n_a)
    bad_toks = ["0"] * (n_listed - 1) + [str(n_a)]
    for toks, fails in ((ok_toks, False), (bad_toks, True)):
        mm = "A+a?" + "".join("," + t for t in toks) + ";C+m?,0,0;"
        rec = pybam.make_record(0, 0, 0, seq, "3000M", mm, [200] * (len(toks) + 2))
        o = O.Oracle(O.parse_mod_codes("m"), [0.8], ["chrT"])
        o.add_contig("chrT", ref.encode())
        if fails:
            with pytest.raises(O.OracleError) as oe:
                o.process(pybam.flatten([good, rec]))
            assert oe.value.code == 10
            with pytest.raises(minimod_amd.MinimodHipError) as he:
                hip_rows_from_records([good, rec, good], ref, "m")
            assert (he.value.code, he.value.read) == (10, 1)
        else:
            o.process(pybam.flatten([good, rec]))
            assert hip_rows_from_records([good, rec], ref, "m") == oracle_rows([good, rec], ref, "m")
        o.close()
