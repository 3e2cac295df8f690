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
