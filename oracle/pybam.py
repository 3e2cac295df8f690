"""Pure-Python BAM reader + batch flattener.  TEST INFRASTRUCTURE ONLY.

This module belongs to the oracle: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The product path has its own C
reader (minimod_amd/csrc/host/bamio.c); the two are cross-checked in tests.

What it restates (reference file:line, all under /root/reference):
  * the htslib record layout the reference consumes through accessors
    (src/mod.c:123-202 aux tags, src/mod.c:776-790 core fields / cigar,
    src/mod.c:956,978 packed sequence),
  * the read filters of load_db (src/minimod.c:249-290): unmapped, secondary
    (unless allow_secondary), supplementary (if skip_supplementary),
    l_qseq == 0, missing MM:Z tag; a read with MM but no valid ML:B:C is kept
    with an empty ML,
  * batch boundaries of load_db (src/minimod.c:249): at most K accepted reads
    or sum(l_data) >= B bytes of accepted records.

BGZF is a series of gzip members, so Python's gzip module decodes it.
"""
import gzip
import struct

import numpy as np

BAM_FUNMAP = 0x4
BAM_FREVERSE = 0x10
BAM_FSECONDARY = 0x100
BAM_FSUPPLEMENTARY = 0x800

# 64-byte per-read record shared (by layout) with include/minimod_hip.h mm_read_t
READ_DTYPE = np.dtype([
    ("cigar_off", "<u8"), ("seq_off", "<u8"), ("mm_off", "<u8"), ("ml_off", "<u8"),
    ("tid", "<i4"), ("pos", "<i4"), ("l_qseq", "<u4"), ("n_cigar", "<u4"),
    ("mm_len", "<u4"), ("ml_len", "<u4"), ("flag", "<u2"), ("hp", "u1"), ("rsvd", "u1"),
    ("rsvd2", "<u4"),
], align=False)
assert READ_DTYPE.itemsize == 64

_AUX_FIXED = {b"A": 1, b"c": 1, b"C": 1, b"s": 2, b"S": 2, b"i": 4, b"I": 4, b"f": 4, b"d": 8}
_AUX_INT = {b"c": "<b", b"C": "<B", b"s": "<h", b"S": "<H", b"i": "<i", b"I": "<I"}
_B_SIZE = {b"c": 1, b"C": 1, b"s": 2, b"S": 2, b"i": 4, b"I": 4, b"f": 4}


def _aux_iter(aux):
    """Yield (tag, type, value_bytes_or_tuple) walking the aux block like bam_aux_get."""
    i, n = 0, len(aux)
    while i + 3 <= n:
        tag = aux[i:i + 2]
        typ = aux[i + 2:i + 3]
        i += 3
        if typ in _AUX_FIXED:
            sz = _AUX_FIXED[typ]
            yield tag, typ, aux[i:i + sz]
            i += sz
        elif typ in (b"Z", b"H"):
            j = aux.index(b"\0", i)
            yield tag, typ, aux[i:j]
            i = j + 1
        elif typ == b"B":
            sub = aux[i:i + 1]
            cnt = struct.unpack_from("<i", aux, i + 1)[0]
            sz = _B_SIZE[sub]
            yield tag, typ, (sub, cnt, aux[i + 5:i + 5 + cnt * sz])
            i += 5 + cnt * sz
        else:
            raise ValueError("bad aux type %r" % typ)


class BamRecord(object):
    __slots__ = ("tid", "pos", "mapq", "flag", "l_qseq", "n_cigar", "qname", "cigar",
                 "seq", "aux", "l_data")

    def first_tag(self, tag):
        for t, typ, val in _aux_iter(self.aux):
            if t == tag:
                return typ, val
        return None

    def mm(self):
        """get_mm_tag_ptr (src/mod.c:123-140): first MM tag, must be Z/H, else None."""
        r = self.first_tag(b"MM")
        if r is None or r[0] not in (b"Z", b"H"):
            return None
        return r[1]

    def ml(self):
        """get_ml_tag (src/mod.c:142-185): B:C array with len>0, else None."""
        r = self.first_tag(b"ML")
        if r is None or r[0] != b"B":
            return None
        sub, cnt, payload = r[1]
        if cnt == 0 or sub != b"C":
            return None
        return payload

    def hp(self):
        """get_hp_tag (src/mod.c:188-202): (uint8) bam_aux2i, 0 when absent."""
        r = self.first_tag(b"HP")
        if r is None:
            return 0
        typ, val = r
        if typ in _AUX_INT:
            return struct.unpack(_AUX_INT[typ], val)[0] & 0xFF
        return 0

    def md(self):
        r = self.first_tag(b"MD")
        return None if r is None else r[1]

    def seq_str(self):
        tab = "=ACMGRSVTWYHKDBN"
        out = []
        for i in range(self.l_qseq):
            b = self.seq[i >> 1]
            out.append(tab[(b >> 4) if (i & 1) == 0 else (b & 15)])
        return "".join(out)


class BamFile(object):
    def __init__(self, path):
        self.fp = gzip.open(path, "rb")
        if self.fp.read(4) != b"BAM\1":
            raise ValueError("not a BAM file: %s" % path)
        l_text = struct.unpack("<i", self.fp.read(4))[0]
        self.text = self.fp.read(l_text)
        n_ref = struct.unpack("<i", self.fp.read(4))[0]
        self.target_name = []
        self.target_len = []
        for _ in range(n_ref):
            l_name = struct.unpack("<i", self.fp.read(4))[0]
            self.target_name.append(self.fp.read(l_name)[:-1].decode())
            self.target_len.append(struct.unpack("<i", self.fp.read(4))[0])

    def __iter__(self):
        return self

    def __next__(self):
        hdr = self.fp.read(4)
        if len(hdr) < 4:
            raise StopIteration
        block_size = struct.unpack("<i", hdr)[0]
        blk = self.fp.read(block_size)
        if len(blk) < block_size:
            raise StopIteration
        (tid, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq,
         _ntid, _npos, _tlen) = struct.unpack_from("<iiBBHHHiiii", blk, 0)
        r = BamRecord()
        r.tid, r.pos, r.mapq, r.flag, r.l_qseq, r.n_cigar = tid, pos, mapq, flag, l_seq, n_cigar
        o = 32
        r.qname = blk[o:o + l_read_name - 1]
        o += l_read_name
        r.cigar = np.frombuffer(blk, dtype="<u4", count=n_cigar, offset=o).copy()
        o += 4 * n_cigar
        r.seq = blk[o:o + (l_seq + 1) // 2]
        o += (l_seq + 1) // 2
        o += l_seq  # qual
        r.aux = blk[o:]
        # htslib l_data = block_size - 32 + qname padding to a multiple of 4
        r.l_data = block_size - 32 + ((4 - (l_read_name & 3)) & 3)
        return r

    def close(self):
        self.fp.close()


def _pad(n, a):
    return (n + a - 1) // a * a


def flatten(records):
    """Flatten accepted records into the SoA batch (reads[], cigar/seq/mm/ml pools)."""
    n = len(records)
    reads = np.zeros(n, dtype=READ_DTYPE)
    cig_parts, seq_parts, mm_parts, ml_parts = [], [], [], []
    co = so = mo = lo = 0
    for i, r in enumerate(records):
        mm = r.mm()
        ml = r.ml() or b""
        rd = reads[i]
        rd["cigar_off"], rd["seq_off"], rd["mm_off"], rd["ml_off"] = co, so, mo, lo
        rd["tid"], rd["pos"], rd["l_qseq"], rd["n_cigar"] = r.tid, r.pos, r.l_qseq, r.n_cigar
        rd["mm_len"], rd["ml_len"], rd["flag"], rd["hp"] = len(mm), len(ml), r.flag, r.hp()
        c = np.zeros(_pad(r.n_cigar, 4), dtype="<u4")
        c[:r.n_cigar] = r.cigar
        cig_parts.append(c)
        co += len(c)
        s = np.zeros(_pad(len(r.seq), 16), dtype="u1")
        s[:len(r.seq)] = np.frombuffer(r.seq, dtype="u1")
        seq_parts.append(s)
        so += len(s)
        m = np.zeros(_pad(len(mm) + 1, 16), dtype="u1")
        m[:len(mm)] = np.frombuffer(mm, dtype="u1")
        mm_parts.append(m)
        mo += len(m)
        l = np.zeros(_pad(len(ml), 4), dtype="u1")
        l[:len(ml)] = np.frombuffer(ml, dtype="u1")
        ml_parts.append(l)
        lo += len(l)
    tail = 64  # zero slack so vector over-reads stay inside the pools
    def cat(parts, dt, slack):
        parts = parts + [np.zeros(slack, dtype=dt)]
        return np.ascontiguousarray(np.concatenate(parts))
    return {
        "reads": reads,
        "cigar": cat(cig_parts, "<u4", tail // 4),
        "seq": cat(seq_parts, "u1", tail),
        "mm": cat(mm_parts, "u1", tail),
        "ml": cat(ml_parts, "u1", tail),
        "qnames": [r.qname for r in records],
    }


def accept(r, allow_secondary=False, skip_supplementary=False):
    """load_db filters, src/minimod.c:260-284."""
    if r.flag & BAM_FUNMAP:
        return False
    if (not allow_secondary) and (r.flag & BAM_FSECONDARY):
        return False
    if skip_supplementary and (r.flag & BAM_FSUPPLEMENTARY):
        return False
    if r.l_qseq == 0:
        return False
    if r.mm() is None:
        return False
    return True


def load_batches(path, K=512, B=20 * 1000 * 1000, allow_secondary=False, skip_supplementary=False):
    """Yield (header, flattened batch, stats) following load_db's -K/-B rule."""
    bam = BamFile(path)
    it = iter(bam)
    done = False
    while not done:
        recs, nbytes, total = [], 0, 0
        while len(recs) < K and nbytes < B:
            try:
                r = next(it)
            except StopIteration:
                done = True
                break
            total += 1
            if not accept(r, allow_secondary, skip_supplementary):
                continue
            recs.append(r)
            nbytes += r.l_data
        yield bam, flatten(recs), {"total_reads": total, "n_reads": len(recs), "bytes": nbytes}
        # freq_main.c:410 loop condition: continue while the last batch was full
        if not (len(recs) >= K or nbytes >= B):
            break
    bam.close()


def load_all(path, **kw):
    """All accepted records of a BAM as one flattened batch (goldens are batch-invariant)."""
    bam = BamFile(path)
    recs = [r for r in bam if accept(r, kw.get("allow_secondary", False), kw.get("skip_supplementary", False))]
    bam.close()
    return bam, flatten(recs)


_CIG_OPS = "MIDNSHP=X"
_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def parse_cigar(s):
    """'5M2I3M' -> uint32 array of len<<4|op."""
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | _CIG_OPS.index(ch))
            num = ""
    return np.array(out, dtype="<u4")


def pack_seq(s):
    b = bytearray((len(s) + 1) // 2)
    for i, ch in enumerate(s):
        v = _NT16.get(ch.upper(), 15)
        b[i >> 1] |= (v << 4) if (i & 1) == 0 else v
    return bytes(b)


def make_record(tid, pos, flag, seq, cigar, mm, ml=None, hp=None, qname=b"r"):
    """Hand-built record (known-answer tests): seq/cigar/mm as text, ml as a list of ints."""
    r = BamRecord()
    r.tid, r.pos, r.mapq, r.flag = tid, pos, 60, flag
    r.l_qseq = len(seq)
    r.cigar = parse_cigar(cigar) if isinstance(cigar, str) else np.asarray(cigar, dtype="<u4")
    r.n_cigar = len(r.cigar)
    r.qname = qname
    r.seq = pack_seq(seq)
    aux = b""
    if mm is not None:
        aux += b"MMZ" + (mm.encode() if isinstance(mm, str) else mm) + b"\0"
    if ml is not None:
        aux += b"MLBC" + struct.pack("<i", len(ml)) + bytes(bytearray(ml))
    if hp is not None:
        aux += b"HPC" + bytes(bytearray([hp]))
    r.aux = aux
    lq = len(qname) + 1
    r.l_data = _pad(lq, 4) + 4 * r.n_cigar + len(r.seq) + r.l_qseq + len(aux)
    return r
