"""ctypes binding of csrc/host/synth.c: synthetic reference + flattened ONT/HiFi-shaped batches."""
import ctypes
import os

import numpy as np

from .build import build_host, lib_path
from .engine import READ_DTYPE, mm_batch_t


class mm_synth_opts_t(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("contig_len", ctypes.c_int64), ("region_begin", ctypes.c_int64),
                ("region_len", ctypes.c_int64), ("n_reads_total", ctypes.c_int64), ("median_len", ctypes.c_double),
                ("max_len", ctypes.c_double), ("dot_fraction", ctypes.c_double), ("tid", ctypes.c_int32),
                ("shape", ctypes.c_int32), ("single_code", ctypes.c_int32), ("haplotypes", ctypes.c_int32),
                ("long_insertions", ctypes.c_int32), ("rsvd", ctypes.c_int32)]


class mm_host_batch_t(ctypes.Structure):
    _fields_ = [("b", mm_batch_t), ("n_bases", ctypes.c_uint64), ("n_listed_calls", ctypes.c_uint64)]


_lib = None


def host_lib():
    global _lib
    if _lib is None:
        path = lib_path("libminimod_host.so")
        if not os.path.exists(path):
            build_host()
        L = ctypes.CDLL(path)
        L.mm_synth_reference.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p]
        L.mm_synth_reference_slice.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        L.mm_synth_batch.argtypes = [ctypes.POINTER(mm_synth_opts_t), ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                     ctypes.POINTER(mm_host_batch_t)]
        L.mm_synth_batch_free.argtypes = [ctypes.POINTER(mm_host_batch_t)]
        L.mm_batch_make_order.argtypes = [ctypes.POINTER(mm_host_batch_t)]
        L.mm_bam_writer_open.restype = ctypes.c_void_p
        L.mm_bam_writer_open.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int64)]
        L.mm_bam_writer_put_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(mm_batch_t), ctypes.c_int]
        L.mm_bam_writer_close.argtypes = [ctypes.c_void_p]
        L.mm_write_fasta.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64]
        _lib = L
    return _lib


def reference(seed, length):
    out = np.empty(int(length), dtype=np.uint8)
    host_lib().mm_synth_reference(int(seed), int(length), out.ctypes.data)
    return out


def reference_slice(seed, begin, length):
    """Positions [begin, begin+length) of the synthetic reference; begin must be a multiple of 1 MiB."""
    assert begin % (1 << 20) == 0
    out = np.empty(int(length), dtype=np.uint8)
    host_lib().mm_synth_reference_slice(int(seed), int(begin), int(length), out.ctypes.data)
    return out


def _view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    nbytes = n * np.dtype(dtype).itemsize
    buf = (ctypes.c_char * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


def batch(ref, first_read, n_reads, seed=0x5EED, contig_len=None, n_reads_total=None, shape=0, tid=0, single_code=False,
          dot_fraction=0.0, haplotypes=False, long_insertions=False, region_begin=0, region_len=0, median_len=0.0,
          max_len=0.0, with_order=True):
    """Generate reads [first_read, first_read+n_reads) as a numpy batch dict (same layout as oracle.pybam.flatten)."""
    o = mm_synth_opts_t()
    o.seed, o.contig_len = int(seed), int(contig_len if contig_len is not None else len(ref))
    o.region_begin, o.region_len = int(region_begin), int(region_len)
    o.n_reads_total = int(n_reads_total if n_reads_total is not None else n_reads)
    o.median_len, o.max_len, o.dot_fraction = float(median_len), float(max_len), float(dot_fraction)
    o.tid, o.shape, o.single_code = int(tid), int(shape), int(single_code)
    o.haplotypes, o.long_insertions = int(haplotypes), int(long_insertions)
    hb = mm_host_batch_t()
    r = host_lib().mm_synth_batch(ctypes.byref(o), ref.ctypes.data, int(first_read), int(n_reads), ctypes.byref(hb))
    if r:
        raise RuntimeError("mm_synth_batch failed")
    if with_order:
        host_lib().mm_batch_make_order(ctypes.byref(hb))
    b = hb.b
    out = {
        "reads": _view(b.reads, b.n_reads, READ_DTYPE),
        "cigar": _view(b.cigar, b.n_cigar_words, np.dtype("<u4")),
        "seq": _view(b.seq, b.n_seq_bytes, np.uint8),
        "mm": _view(b.mm, b.n_mm_bytes, np.uint8),
        "ml": _view(b.ml, b.n_ml_bytes, np.uint8),
        "order": _view(b.order, b.n_reads, np.dtype("<i4")) if with_order and b.order else None,
        "n_bases": int(hb.n_bases), "n_listed_calls": int(hb.n_listed_calls),
        "max_n_cigar": int(b.max_n_cigar), "max_l_qseq": int(b.max_l_qseq),
    }
    host_lib().mm_synth_batch_free(ctypes.byref(hb))
    return out


def write_bam(path, contigs, batches, filter_fodder=True):
    """Write numpy batches (dicts as returned by batch()) as one coordinate-sorted BGZF BAM.  contigs: [(name, length)]."""
    from .engine import batch_struct
    L = host_lib()
    names = (ctypes.c_char_p * len(contigs))(*[n.encode() for n, _ in contigs])
    lens = (ctypes.c_int64 * len(contigs))(*[int(l) for _, l in contigs])
    w = L.mm_bam_writer_open(path.encode(), len(contigs), names, lens)
    if not w:
        raise IOError("cannot create %s" % path)
    for b in batches:
        bs = batch_struct(b)
        if L.mm_bam_writer_put_batch(w, ctypes.byref(bs), int(filter_fodder)):
            raise IOError("write failed")
    if L.mm_bam_writer_close(w):
        raise IOError("close failed")


def write_fasta(path, name, seq):
    seq = np.ascontiguousarray(seq)
    if host_lib().mm_write_fasta(path.encode(), name.encode(), seq.ctypes.data, len(seq)):
        raise IOError("cannot write %s" % path)
