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
        L.mm_bam_writer_open_piece.restype = ctypes.c_void_p
        L.mm_bam_writer_open_piece.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int64),
                                               ctypes.c_int, ctypes.c_uint64]
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


def concat(batches):
    """Several numpy batches as one (reads in the order given): pools joined without the 64-byte slack between them,
    read offsets shifted.  A coordinate-sorted BAM over several contigs gives batches like this: reads of more than one
    contig in one -K batch."""
    batches = [b for b in batches if len(b["reads"])]
    if not batches:
        raise ValueError("nothing to concatenate")
    reads, cig, seq, mm, ml = [], [], [], [], []
    oc = os_ = om = ol = 0
    for b in batches:
        r = b["reads"].copy()
        r["cigar_off"] += oc; r["seq_off"] += os_; r["mm_off"] += om; r["ml_off"] += ol
        reads.append(r)
        c, q, m, l = b["cigar"][:len(b["cigar"]) - 16], b["seq"][:len(b["seq"]) - 64], b["mm"][:len(b["mm"]) - 64], b["ml"][:len(b["ml"]) - 64]
        assert len(c) % 4 == 0 and len(q) % 16 == 0 and len(m) % 16 == 0 and len(l) % 4 == 0
        cig.append(c); seq.append(q); mm.append(m); ml.append(l)
        oc += len(c); os_ += len(q); om += len(m); ol += len(l)
    z = lambda n, dt: np.zeros(n, dtype=dt)
    rd = np.concatenate(reads)
    return {"reads": rd, "cigar": np.concatenate(cig + [z(16, "<u4")]), "seq": np.concatenate(seq + [z(64, np.uint8)]),
            "mm": np.concatenate(mm + [z(64, np.uint8)]), "ml": np.concatenate(ml + [z(64, np.uint8)]), "order": None,
            "n_bases": int(sum(b["n_bases"] for b in batches)), "n_listed_calls": int(sum(b["n_listed_calls"] for b in batches)),
            "max_n_cigar": int(rd["n_cigar"].max()), "max_l_qseq": int(rd["l_qseq"].max())}


def split(batch, sizes):
    """The reads of a numpy batch cut into consecutive batches of the given sizes (pools shared, offsets unchanged)."""
    out, lo = [], 0
    for n in sizes:
        b = dict(batch)
        b["reads"] = np.ascontiguousarray(batch["reads"][lo:lo + n])
        b["order"] = None
        b["n_bases"] = int(b["reads"]["l_qseq"].sum())
        b["max_n_cigar"] = int(b["reads"]["n_cigar"].max()) if n else 0
        b["max_l_qseq"] = int(b["reads"]["l_qseq"].max()) if n else 0
        out.append(b)
        lo += n
    assert lo == len(batch["reads"])
    return out


def multi_contig(refs, reads_per_contig, batch_reads, seed=0x5EED, **kw):
    """A coordinate-sorted read set over several contigs as -K sized batches.  refs: [reference array or None by tid]
    (None = the contig carries no reads); reads_per_contig: reads on each.  Batches hold `batch_reads` reads and run
    across contig boundaries like load_db's do on a sorted BAM."""
    parts = []
    for tid, (ref, n) in enumerate(zip(refs, reads_per_contig)):
        if ref is None or n == 0:
            continue
        parts.append(batch(ref, 0, n, seed=seed + 101 * tid, n_reads_total=n, tid=tid, with_order=False, **kw))
    whole = concat(parts)
    n = len(whole["reads"])
    sizes = [min(batch_reads, n - i) for i in range(0, n, batch_reads)]
    return split(whole, sizes)


def write_fasta_multi(path, contigs):
    """contigs: [(name, sequence array)] as one FASTA file (80 columns)."""
    with open(path, "wb") as f:
        for name, seq in contigs:
            f.write(b">" + name.encode() + b" synthetic\n")
            a = np.ascontiguousarray(seq, dtype=np.uint8)
            n = len(a) // 80 * 80
            if n:
                body = np.empty((n // 80, 81), dtype=np.uint8)
                body[:, :80] = a[:n].reshape(-1, 80)
                body[:, 80] = 10
                f.write(body.tobytes())
            if n < len(a):
                f.write(a[n:].tobytes() + b"\n")


def write_bam(path, contigs, batches, filter_fodder=True, index=False):
    """Write numpy batches (dicts as returned by batch()) as one coordinate-sorted BGZF BAM.  contigs: [(name, length)].
    index=True also writes path + ".bai" (bins, chunks and the 16 kb linear index of the SAM specification)."""
    from .engine import batch_struct
    L = host_lib()
    names = (ctypes.c_char_p * len(contigs))(*[n.encode() for n, _ in contigs])
    lens = (ctypes.c_int64 * len(contigs))(*[int(l) for _, l in contigs])
    w = L.mm_bam_writer_open_piece(path.encode(), len(contigs), names, lens, 4 if index else 0, 0)
    if not w:
        raise IOError("cannot create %s" % path)
    for b in batches:
        bs = batch_struct(b)
        if L.mm_bam_writer_put_batch(w, ctypes.byref(bs), int(filter_fodder)):
            raise IOError("write failed")
    if L.mm_bam_writer_close(w):
        raise IOError("close failed")


def write_bam_parallel(path, contigs, batches, filter_fodder=True, threads=8):
    """write_bam with one piece per batch deflated on `threads` threads and the pieces concatenated: the same bytes as
    write_bam would give when every batch ends its BGZF block (it does not: blocks run on across batches there), the same
    RECORDS in any case."""
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    from .engine import batch_struct
    L = host_lib()
    names = (ctypes.c_char_p * len(contigs))(*[n.encode() for n, _ in contigs])
    lens = (ctypes.c_int64 * len(contigs))(*[int(l) for _, l in contigs])
    firsts, acc = [], 0
    for b in batches:
        firsts.append(acc)
        acc += len(b["reads"])
    nb = len(batches)

    def piece(i):
        flags = (1 if i > 0 else 0) | (2 if i < nb - 1 else 0)
        pp = "%s.piece%d" % (path, i)
        w = L.mm_bam_writer_open_piece(pp.encode(), len(contigs), names, lens, flags, firsts[i])
        if not w:
            raise IOError("cannot create %s" % pp)
        bs = batch_struct(batches[i])
        if L.mm_bam_writer_put_batch(w, ctypes.byref(bs), int(filter_fodder)) or L.mm_bam_writer_close(w):
            raise IOError("write failed")
        return pp
    if nb == 0:
        return write_bam(path, contigs, batches, filter_fodder)
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        pieces = list(ex.map(piece, range(nb)))
    with open(path, "wb") as out:
        for pp in pieces:
            with open(pp, "rb") as f:
                shutil.copyfileobj(f, out, 1 << 24)
            os.remove(pp)


def read_bai(path):
    """A .bai as [(bins, lin)] per reference + n_no_coor: bins = {bin: [(beg, end), ...]}, lin = the linear index (SAM spec 5.2)."""
    import struct
    d = open(path, "rb").read()
    if d[:4] != b"BAI\1":
        raise ValueError("not a .bai: %s" % path)
    n_ref, = struct.unpack_from("<i", d, 4)
    o, refs = 8, []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", d, o); o += 4
        bins = {}
        for _ in range(n_bin):
            b_, n_chunk = struct.unpack_from("<Ii", d, o); o += 8
            v = struct.unpack_from("<%dQ" % (2 * n_chunk), d, o); o += 16 * n_chunk
            bins[b_] = [(v[2 * i], v[2 * i + 1]) for i in range(n_chunk)]
        n_intv, = struct.unpack_from("<i", d, o); o += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, d, o)); o += 8 * n_intv
        refs.append((bins, lin))
    n_no_coor = struct.unpack_from("<Q", d, o)[0] if o + 8 <= len(d) else 0
    return refs, n_no_coor


def merge_bai(out_path, parts):
    """The index of a BAM that is pieces one behind the other: parts = [(piece's .bai as read_bai gives it, the piece's first byte in
    the whole file)] in file order (every piece begins a BGZF block, so a virtual offset moves by first_byte << 16).  Chunks of a bin
    that continue each other across a joint are joined; a 16 kb window takes the first piece's entry that has one."""
    import struct
    n_ref = len(parts[0][0][0])
    out = [b"BAI\1", struct.pack("<i", n_ref)]
    no_coor = sum(p[0][1] for p in parts)
    for t in range(n_ref):
        bins, lin = {}, []
        for (refs, _), base in parts:
            sh = base << 16
            pb, pl = refs[t]
            for b_, chunks in pb.items():
                dst = bins.setdefault(b_, [])
                for beg, end in chunks:
                    if dst and dst[-1][1] == beg + sh:
                        dst[-1] = (dst[-1][0], end + sh)
                    else:
                        dst.append((beg + sh, end + sh))
            # a piece's windows: [0, first record's window) are 0 = nothing there; inside, the writer has filled every window.  The
            # first piece that reaches a window decides it (records are sorted over the pieces: no later piece has an earlier record there)
            if len(pl) > len(lin):
                lin.extend((v + sh) if (v or base) else 0 for v in pl[len(lin):])
        out.append(struct.pack("<i", len(bins)))
        for b_ in sorted(bins):
            ch = bins[b_]
            out.append(struct.pack("<Ii", b_, len(ch)))
            out.append(struct.pack("<%dQ" % (2 * len(ch)), *[x for c in ch for x in c]))
        out.append(struct.pack("<i", len(lin)))
        out.append(struct.pack("<%dQ" % len(lin), *lin))
    out.append(struct.pack("<Q", no_coor))
    with open(out_path, "wb") as f:
        f.write(b"".join(out))


def write_bam_rounds(path, contigs, batches, first_round=True, last_round=True, first_read=0, filter_fodder=True, threads=8, index=False):
    """One ROUND of a BAM written in several rounds (a file too big to hold as batches at once): like write_bam_parallel, with the
    header only in the first round's first piece and the EOF block only behind the last round's last piece; the rounds' files
    concatenated are the BAM.  index=True: returns [(piece's index, piece's first byte inside this round's file)] for merge_bai."""
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    from .engine import batch_struct
    L = host_lib()
    names = (ctypes.c_char_p * len(contigs))(*[n.encode() for n, _ in contigs])
    lens = (ctypes.c_int64 * len(contigs))(*[int(l) for _, l in contigs])
    firsts, acc = [], first_read
    for b in batches:
        firsts.append(acc)
        acc += len(b["reads"])
    nb = len(batches)

    def piece(i):
        flags = (0 if (first_round and i == 0) else 1) | (0 if (last_round and i == nb - 1) else 2) | (4 if index else 0)
        pp = "%s.piece%d" % (path, i)
        w = L.mm_bam_writer_open_piece(pp.encode(), len(contigs), names, lens, flags, firsts[i])
        if not w:
            raise IOError("cannot create %s" % pp)
        bs = batch_struct(batches[i])
        if L.mm_bam_writer_put_batch(w, ctypes.byref(bs), int(filter_fodder)) or L.mm_bam_writer_close(w):
            raise IOError("write failed")
        return pp
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        pieces = list(ex.map(piece, range(nb)))
    parts, at = [], 0
    with open(path, "wb") as out:
        for pp in pieces:
            if index:
                parts.append((read_bai(pp + ".bai"), at))
                os.remove(pp + ".bai")
                at += os.path.getsize(pp)
            with open(pp, "rb") as f:
                shutil.copyfileobj(f, out, 1 << 24)
            os.remove(pp)
    return parts


def write_fasta(path, name, seq):
    seq = np.ascontiguousarray(seq)
    if host_lib().mm_write_fasta(path.encode(), name.encode(), seq.ctypes.data, len(seq)):
        raise IOError("cannot write %s" % path)
