"""Random differential run of the ERROR paths: one read of a mixed batch is corrupted (MM text, ML length, CIGAR, position),
the oracle's (error code, read index) must be the HIP path's in every mode -- or both must finish with the same rows.
usage: tools/fuzz_errors.py <first> <count>"""
import time, struct, copy
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minimod_amd
from oracle import pybam, oracle as O
from tests import test_hip_stream_gpu as T

def aux_parts(r):
    a = r.aux; z = a.index(b"\0")
    mm = a[3:z].decode(); rest = a[z + 1:]
    assert rest[:4] == b"MLBC"
    n = struct.unpack("<i", rest[4:8])[0]
    ml = list(rest[8:8 + n]); tail = rest[8 + n:]
    return mm, ml, tail

def with_aux(r, mm, ml, tail=b""):
    q = copy.copy(r)
    q.aux = b"MMZ" + mm.encode() + b"\0" + b"MLBC" + struct.pack("<i", len(ml)) + bytes(bytearray(ml)) + tail
    return q

def corrupt(rng, r, ref_len):
    mm, ml, tail = aux_parts(r)
    kind = int(rng.integers(0, 12))
    if kind == 0 and len(mm) > 6:      # a character that is not a digit inside the list
        i = int(rng.integers(5, len(mm))); mm = mm[:i] + str(rng.choice(list("x-+ a"))) + mm[i + 1:]
    elif kind == 1 and len(mm) > 6:    # a token far too long
        i = mm.find(",")
        if i > 0: mm = mm[:i + 1] + "1" * int(rng.integers(9, 14)) + mm[i + 1:]
    elif kind == 2:                    # a rank past the read's last base
        mm = mm[:-1] + ",%d;" % int(rng.integers(r.l_qseq, 3 * r.l_qseq + 10))
        ml = ml + [1, 2, 3, 4]
    elif kind == 3 and ml:             # ML too short
        ml = ml[:int(rng.integers(0, len(ml)))]
    elif kind == 4:                    # no closing semicolon
        mm = mm[:-1]
    elif kind == 5:                    # a header that is not one
        mm = str(rng.choice(["C", "C+", "+m?,1;", "Cm?,1;", "C+m?1;", "X+m?,1;", "C+mmmmmmmmmmmmmmmmmm,1;", "C*m,1;"])) + mm
    elif kind == 6:                    # an unknown CIGAR op / hard clip
        q = copy.copy(r); cg = r.cigar.copy(); cg[int(rng.integers(0, len(cg)))] = (5 << 4) | int(rng.choice([5, 6, 9]))
        q.cigar = cg; return q
    elif kind == 7:                    # the CIGAR consumes more than the read has
        q = copy.copy(r); cg = r.cigar.copy(); cg[0] = ((int(cg[0]) >> 4) + r.l_qseq) << 4 | (int(cg[0]) & 15); q.cigar = cg; return q
    elif kind == 8:                    # alignment beyond the contig
        q = copy.copy(r); q.pos = ref_len - 10; return q
    elif kind == 9:                    # empty tokens
        mm = mm.replace(",", ",,", 1)
    elif kind == 10:                   # strand '-' / several flags
        mm = mm.replace("+", "-", 1)
    else:                              # numeric code / ChEBI
        mm = mm.replace("+m", "+21839", 1) if "+m" in mm else mm
    return with_aux(r, mm, ml, tail)

first, count = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = 0; n_err = 0; known = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    ref = T.make_ref(rng, 120000)
    recs = [T._mixed_read(rng, ref) for _ in range(int(rng.integers(5, 60)))]
    k = int(rng.integers(0, len(recs)))
    recs[k] = corrupt(rng, recs[k], len(ref))
    c = ("m", "m,h", "m[*],a[*]")[int(rng.integers(0, 3))]
    try:
        want = ("rows", T.oracle_rows(recs, ref, c))
    except O.OracleError as e:
        want = ("error", e.code, e.read); n_err += 1
    for kw in (dict(stream_mode=3), dict(stream_mode=2), dict(stream_mode=1), dict(force_fused=True)):
        try:
            got = ("rows", T.hip_rows(recs, ref, c, **kw)[0])
        except minimod_amd.MinimodHipError as e:
            got = ("error", e.code, e.read)
        if got != want:
            # the two deviations DESIGN.md section 7 lists: more than 16 code letters in a group; a reverse read whose CIGAR
            # overshoots the sequence in its leading soft clip / insertion
            if "mmmmmmmmmmmmmmmmmm" in aux_parts(recs[k])[0] and got[0] == "error" and got[1] == 5:
                known += 1; print("KNOWN (code letters) seed", seed, "read", k, c, kw, "MM", aux_parts(recs[k])[0][:90], "oracle", str(want)[:60], flush=True); continue
            if got[0] == "error" and got[1] == 14 and (recs[k].flag & 16) and want[0] == "rows":
                known += 1; print("KNOWN (query overrun) seed", seed, "read", k, c, kw, "oracle", str(want)[:40], flush=True); continue
            bad += 1
            print("MISMATCH seed", seed, "read", k, c, kw, "hip", str(got)[:80], "oracle", str(want)[:80], "| MM", aux_parts(recs[k])[0][:60], flush=True)
print("seeds %d..%d done in %.0f s, %d problems, %d known deviations (%d of the batches fail in the oracle)" % (first, first + count - 1, time.time() - t0, bad, known, n_err))
