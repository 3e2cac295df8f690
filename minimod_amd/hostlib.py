"""ctypes view of the C host library (csrc/host): option parsing, FASTA, load_db flattening.  Used by the tests to
cross-check the C host side against the oracle's independent Python restatements."""
import ctypes

import numpy as np

from .engine import MM_CODE_LEN, MM_MAX_MODS, READ_DTYPE, mm_batch_t
from .synth import host_lib


class mmh_mods_t(ctypes.Structure):
    _fields_ = [("n_mods", ctypes.c_int), ("code", (ctypes.c_char * MM_CODE_LEN) * MM_MAX_MODS),
                ("context", (ctypes.c_char * MM_CODE_LEN) * MM_MAX_MODS), ("thresh", ctypes.c_double * MM_MAX_MODS)]


class mmh_ref_t(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int), ("name", ctypes.POINTER(ctypes.c_char_p)),
                ("seq", ctypes.POINTER(ctypes.c_void_p)), ("len", ctypes.POINTER(ctypes.c_int64))]


def _lib():
    L = host_lib()
    if not getattr(L, "_mmh_ready", False):
        L.mmh_parse_mod_codes.argtypes = [ctypes.c_char_p, ctypes.POINTER(mmh_mods_t), ctypes.c_char_p, ctypes.c_size_t]
        L.mmh_parse_mod_threshes.argtypes = [ctypes.c_char_p, ctypes.POINTER(mmh_mods_t), ctypes.c_char_p, ctypes.c_size_t]
        L.mmh_klass_lut.argtypes = [ctypes.c_double, ctypes.c_void_p]
        L.mmh_parse_num.restype = ctypes.c_int64
        L.mmh_parse_num.argtypes = [ctypes.c_char_p]
        L.mmh_load_ref.restype = ctypes.POINTER(mmh_ref_t)
        L.mmh_load_ref.argtypes = [ctypes.c_char_p]
        L.mmh_free_ref.argtypes = [ctypes.POINTER(mmh_ref_t)]
        L.mmh_loader_open.restype = ctypes.c_void_p
        L.mmh_loader_open.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
        L.mmh_loader_next.restype = ctypes.c_int32
        L.mmh_loader_next.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(mm_batch_t), ctypes.POINTER(ctypes.c_int)]
        L.mmh_loader_close.argtypes = [ctypes.c_void_p]
        ctypes.c_int.in_dll(L, "mmh_log_level").value = 1
        L._mmh_ready = True
    return L


def parse_mods(c, m=None):
    """(-c, -m) -> [(code, context, thresh)] through the C parser; raises ValueError with the reference's message."""
    L = _lib()
    mods = mmh_mods_t()
    err = ctypes.create_string_buffer(512)
    if L.mmh_parse_mod_codes((c or "m").encode(), ctypes.byref(mods), err, 512):
        raise ValueError(err.value.decode())
    if not m:
        m = ",".join(["0.8"] * mods.n_mods)
    if L.mmh_parse_mod_threshes(m.encode(), ctypes.byref(mods), err, 512):
        raise ValueError(err.value.decode())
    return [(mods.code[i].value.decode(), mods.context[i].value.decode(), mods.thresh[i]) for i in range(mods.n_mods)]


def klass_lut(th):
    out = np.zeros(256, dtype=np.uint8)
    _lib().mmh_klass_lut(float(th), out.ctypes.data)
    return out


def parse_num(s):
    return int(_lib().mmh_parse_num(s.encode()))


def load_ref(path, threads=1):
    """threads >= 1: a plain file through the mapped parser with that many workers; 0: the stream parser (what .gz files get)"""
    L = _lib()
    L.mmh_load_ref_mt.restype = ctypes.POINTER(mmh_ref_t)
    L.mmh_load_ref_mt.argtypes = [ctypes.c_char_p, ctypes.c_int]
    r = L.mmh_load_ref_mt(path.encode(), int(threads))
    if not r:
        raise IOError("cannot open %s" % path)
    out = []
    for i in range(r.contents.n):
        n = r.contents.len[i]
        buf = (ctypes.c_char * n).from_address(r.contents.seq[i]) if n else b""
        out.append((r.contents.name[i].decode(), bytes(buf)))
    L.mmh_free_ref(r)
    return out


def load_batches(path, K=512, B=20 * 1000 * 1000, threads=2, allow_secondary=False, skip_supplementary=False):
    """Yield numpy batch dicts produced by the C loader (copies)."""
    L = _lib()
    ld = L.mmh_loader_open(path.encode(), threads, K, B, int(allow_secondary), int(skip_supplementary))
    if not ld:
        raise IOError("cannot open %s" % path)
    more = ctypes.c_int(1)
    s = 0
    try:
        while more.value:
            b = mm_batch_t()
            n = L.mmh_loader_next(ld, s, ctypes.byref(b), ctypes.byref(more))
            if n < 0:
                raise IOError("corrupt BAM %s" % path)

            def view(ptr, count, dt):
                if count == 0:
                    return np.zeros(0, dtype=dt)
                nb = count * np.dtype(dt).itemsize
                return np.frombuffer((ctypes.c_char * nb).from_address(ptr), dtype=dt).copy()
            yield {"reads": view(b.reads, b.n_reads, READ_DTYPE), "cigar": view(b.cigar, b.n_cigar_words, "<u4"),
                   "seq": view(b.seq, b.n_seq_bytes, np.uint8), "mm": view(b.mm, b.n_mm_bytes, np.uint8),
                   "ml": view(b.ml, b.n_ml_bytes, np.uint8), "max_n_cigar": b.max_n_cigar, "max_l_qseq": b.max_l_qseq}
            s ^= 1
    finally:
        L.mmh_loader_close(ld)


class mm_bam_hdr_t(ctypes.Structure):
    _fields_ = [("n_targets", ctypes.c_int32), ("target_name", ctypes.POINTER(ctypes.c_char_p)),
                ("target_len", ctypes.POINTER(ctypes.c_uint32))]


def format_freq_rows(rows, names, code_names, path, threads=1, bedmethyl=False, insertions=False, haplotypes=False):
    """print_freq_output through the C emitter (rows: engine.ROW_DTYPE) into the file `path`; threads > 1 formats on a
    worker pool.  The header is written too, like the CLI does."""
    from .engine import ROW_DTYPE
    L = _lib()
    libc = ctypes.CDLL(None)
    libc.fopen.restype = ctypes.c_void_p
    libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [ctypes.c_void_p]
    L.mm_pool_create.restype = ctypes.c_void_p
    L.mm_pool_create.argtypes = [ctypes.c_int]
    L.mm_pool_destroy.argtypes = [ctypes.c_void_p]
    L.mmh_print_freq_header.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.mmh_print_freq_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(mm_bam_hdr_t),
                                      ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
    tn = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
    tl = (ctypes.c_uint32 * len(names))(*([0] * len(names)))
    hdr = mm_bam_hdr_t(len(names), tn, tl)
    cn = (ctypes.c_char_p * len(code_names))(*[c.encode() for c in code_names])
    fp = libc.fopen(path.encode(), b"wb")
    if not fp:
        raise OSError("cannot open " + path)
    pool = L.mm_pool_create(threads) if threads > 1 else None
    try:
        L.mmh_print_freq_header(fp, int(bedmethyl), int(insertions), int(haplotypes))
        L.mmh_print_freq_rows(fp, pool, rows.ctypes.data, len(rows), ctypes.byref(hdr), cn, len(code_names), int(bedmethyl),
                              int(insertions), int(haplotypes))
        if L.mmh_emit_finish() != 0:
            raise OSError("write failed")
    finally:
        libc.fclose(fp)
        if pool:
            L.mm_pool_destroy(pool)


def inflate_raw(data, out_len):
    """The BGZF reader's own raw-DEFLATE decoder (inflate_fast.c): bytes -> bytes of exactly out_len, or None if it
    rejects the stream."""
    L = _lib()
    L.mm_inflate_raw.restype = ctypes.c_int
    L.mm_inflate_raw.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    out = ctypes.create_string_buffer(out_len + 16)
    r = L.mm_inflate_raw(bytes(data), len(data), out, out_len)
    return out.raw[:out_len] if r == 0 else None


class mmh_devloader_opts_t(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("n_targets", ctypes.c_int), ("allow_secondary", ctypes.c_int), ("skip_supplementary", ctypes.c_int),
                ("header_bytes", ctypes.c_uint64), ("voffset", ctypes.c_uint64),
                ("ranged", ctypes.c_int), ("first", ctypes.c_int), ("last", ctypes.c_int), ("range_done_before_start", ctypes.c_int),
                ("lo_tid", ctypes.c_int32), ("hi_tid", ctypes.c_int32), ("lo_pos", ctypes.c_int64), ("hi_pos", ctypes.c_int64),
                ("target_bases", ctypes.c_uint64),
                ("group_slots", ctypes.c_int), ("max_blocks", ctypes.c_int), ("arenas", ctypes.c_int),
                ("max_cbytes", ctypes.c_uint64), ("arena_bytes", ctypes.c_uint64), ("head_room", ctypes.c_uint64), ("names", ctypes.c_int)]


class mmh_devbatch_t(ctypes.Structure):
    _fields_ = [("batch", mm_batch_t), ("arena", ctypes.c_int), ("bases", ctypes.c_uint64),
                ("total_reads", ctypes.c_uint64), ("total_bytes", ctypes.c_uint64), ("processed_bytes", ctypes.c_uint64),
                ("names", ctypes.c_void_p), ("name_off", ctypes.c_void_p), ("names_bytes", ctypes.c_uint64)]


class mmh_devloader_stats_t(ctypes.Structure):
    _fields_ = [("total_reads", ctypes.c_uint64), ("total_bytes", ctypes.c_uint64), ("processed_reads", ctypes.c_uint64),
                ("processed_bytes", ctypes.c_uint64), ("processed_bases", ctypes.c_uint64),
                ("groups", ctypes.c_uint64), ("slow_blocks", ctypes.c_uint64), ("patched_blocks", ctypes.c_uint64),
                ("wait_seconds", ctypes.c_double), ("stage_seconds", ctypes.c_double), ("stage_ms", ctypes.c_double * 4), ("err", ctypes.c_int)]


def peek_header(path):
    """(contig names, lengths, offset of the first record in the decoded stream) through mm_bam_peek_header2"""
    L = _lib()
    L.mm_bam_peek_header2.argtypes = [ctypes.c_char_p, ctypes.POINTER(mm_bam_hdr_t), ctypes.POINTER(ctypes.c_uint64)]
    L.mm_bam_hdr_free.argtypes = [ctypes.POINTER(mm_bam_hdr_t)]
    hdr = mm_bam_hdr_t()
    hb = ctypes.c_uint64(0)
    if L.mm_bam_peek_header2(path.encode(), ctypes.byref(hdr), ctypes.byref(hb)) != 0:
        raise IOError("not a BAM file: %s" % path)
    names = [hdr.target_name[i].decode() for i in range(hdr.n_targets)]
    lens = [int(hdr.target_len[i]) for i in range(hdr.n_targets)]
    L.mm_bam_hdr_free(ctypes.byref(hdr))
    return names, lens, int(hb.value)


def load_batches_device(path, threads=2, allow_secondary=False, skip_supplementary=False, target_bases=0, device=0, sizes=None, share=None, voffset=0, names=False, codes=False):
    """Yield (numpy batch dict, per-batch totals) made by the DEVICE loader (devloader.c on include/minimod_ingest.h), copied back to
    the host for comparison with load_batches().  sizes: dict of the small test geometries (group_slots, max_blocks, arenas,
    max_cbytes, arena_bytes, head_room).  share: (lo_tid, lo_pos, hi_tid, hi_pos, first, last).  names: the batch dict also holds "names"
    (the reads' names, a list of bytes); codes: ... and "codes" (mm_ingest_batch_codes' answer: a list of bytes, or the negative code).
    The last item yielded is the stats structure."""
    L = _lib()
    L.mm_pool_create.restype = ctypes.c_void_p
    L.mm_pool_create.argtypes = [ctypes.c_int]
    L.mm_pool_destroy.argtypes = [ctypes.c_void_p]
    L.mmh_devloader_open.restype = ctypes.c_void_p
    L.mmh_devloader_open.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.POINTER(mmh_devloader_opts_t), ctypes.c_char_p, ctypes.c_size_t]
    L.mmh_devloader_next.restype = ctypes.c_int32
    L.mmh_devloader_next.argtypes = [ctypes.c_void_p, ctypes.POINTER(mmh_devbatch_t), ctypes.POINTER(ctypes.c_int)]
    L.mmh_devloader_release.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.mmh_devloader_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.mmh_devloader_stats.restype = ctypes.POINTER(mmh_devloader_stats_t)
    L.mmh_devloader_stats.argtypes = [ctypes.c_void_p]
    L.mmh_devloader_close.argtypes = [ctypes.c_void_p]
    tnames, lens, hb = peek_header(path)   # (the contigs' names: `names` is the caller's switch for the READS' names)
    o = mmh_devloader_opts_t()
    o.device = device; o.n_targets = len(tnames); o.allow_secondary = int(allow_secondary); o.skip_supplementary = int(skip_supplementary)
    o.header_bytes = hb; o.voffset = voffset; o.target_bases = target_bases
    o.names = int(bool(names))
    L.mmh_devloader_codes.restype = ctypes.c_int
    L.mmh_devloader_codes.argtypes = [ctypes.c_void_p, ctypes.POINTER(mm_batch_t), ctypes.c_char_p, ctypes.c_int]
    if share:
        o.ranged = 1
        o.lo_tid, o.lo_pos, o.hi_tid, o.hi_pos, o.first, o.last = share
    for k, v in (sizes or {}).items():
        setattr(o, k, v)
    pool = L.mm_pool_create(threads)
    err = ctypes.create_string_buffer(512)
    dl = L.mmh_devloader_open(path.encode(), pool, ctypes.byref(o), err, 512)
    if not dl:
        L.mm_pool_destroy(pool)
        raise IOError("device loader: " + err.value.decode())
    more = ctypes.c_int(1)
    try:
        while more.value:
            db = mmh_devbatch_t()
            n = L.mmh_devloader_next(dl, ctypes.byref(db), ctypes.byref(more))
            if n < 0:
                raise IOError("corrupt BAM %s (device loader, error %d)" % (path, L.mmh_devloader_stats(dl).contents.err))
            b = db.batch

            def fetch(ptr, count, dt):
                out = np.zeros(count, dtype=dt)
                if count and L.mmh_devloader_fetch(dl, out.ctypes.data, ptr, out.nbytes) != 0:
                    raise IOError("device -> host copy failed")
                return out
            if n > 0:
                d = {"reads": fetch(b.reads, b.n_reads, READ_DTYPE), "cigar": fetch(b.cigar, b.n_cigar_words, "<u4"),
                     "seq": fetch(b.seq, b.n_seq_bytes, np.uint8), "mm": fetch(b.mm, b.n_mm_bytes, np.uint8),
                     "ml": fetch(b.ml, b.n_ml_bytes, np.uint8), "max_n_cigar": b.max_n_cigar, "max_l_qseq": b.max_l_qseq}
                if not names:   # (the default reader geometry -- what a plain freq run uses -- keeps no names)
                    assert not db.names and db.names_bytes == 0, "a reader opened without names handed names out"
                if names:
                    off = fetch(db.name_off, b.n_reads, "<u8")
                    txt = fetch(db.names, db.names_bytes, np.uint8).tobytes()
                    d["names"] = [txt[int(a):txt.index(b"\0", int(a))] for a in off]
                if codes:
                    buf = ctypes.create_string_buffer(16 * 256)
                    k = L.mmh_devloader_codes(dl, ctypes.byref(b), buf, 256)
                    d["codes"] = k if k < 0 else [buf.raw[16 * i:16 * i + 16].split(b"\0")[0] for i in range(k)]
                L.mmh_devloader_release(dl, db.arena)
            else:
                d = None
            yield d, {"bases": int(db.bases), "total_reads": int(db.total_reads), "total_bytes": int(db.total_bytes), "processed_bytes": int(db.processed_bytes)}
        st = L.mmh_devloader_stats(dl).contents
        yield None, {k: (list(getattr(st, k)) if k == "stage_ms" else getattr(st, k)) for k, _ in mmh_devloader_stats_t._fields_}
    finally:
        L.mmh_devloader_close(dl)
        L.mm_pool_destroy(pool)
