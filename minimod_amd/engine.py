"""ctypes mirror of include/minimod_hip.h (the drop-in boundary).  Argument meaning and error behaviour follow the
reference seams cited in that header: create = load_ref + load_ref_contexts + init_core, submit = process_db,
finalize = output_core."""
import ctypes
import importlib.util
import os
import sys

import numpy as np

from .build import build_hip, lib_path

MM_ABI_VERSION = 6
MM_MAX_MODS = 32
MM_CODE_LEN = 16

READ_DTYPE = np.dtype([
    ("cigar_off", "<u8"), ("seq_off", "<u8"), ("mm_off", "<u8"), ("ml_off", "<u8"),
    ("tid", "<i4"), ("pos", "<i4"), ("l_qseq", "<u4"), ("n_cigar", "<u4"),
    ("mm_len", "<u4"), ("ml_len", "<u4"), ("flag", "<u2"), ("hp", "u1"), ("rsvd", "u1"),
    ("rsvd2", "<u4"),
])
ROW_DTYPE = np.dtype([("tid", "<i4"), ("pos", "<i4"), ("strand", "u1"), ("rsvd", "u1"), ("ins_offset", "<u2"),
                      ("code", "<i2"), ("hp", "<i2"), ("n_called", "<u4"), ("n_mod", "<u4")])
VIEW_ROW_DTYPE = np.dtype([("read", "<u4"), ("pos", "<i4"), ("read_pos", "<u4"), ("ins_offset", "<u2"), ("code", "u1"),
                           ("prob", "u1")])
assert READ_DTYPE.itemsize == 64 and ROW_DTYPE.itemsize == 24 and VIEW_ROW_DTYPE.itemsize == 16


class mm_batch_t(ctypes.Structure):
    _fields_ = [("reads", ctypes.c_void_p), ("cigar", ctypes.c_void_p), ("seq", ctypes.c_void_p),
                ("mm", ctypes.c_void_p), ("ml", ctypes.c_void_p), ("order", ctypes.c_void_p),
                ("n_reads", ctypes.c_int32), ("n_order", ctypes.c_int32),
                ("n_cigar_words", ctypes.c_uint64), ("n_seq_bytes", ctypes.c_uint64),
                ("n_mm_bytes", ctypes.c_uint64), ("n_ml_bytes", ctypes.c_uint64),
                ("max_n_cigar", ctypes.c_uint32), ("max_l_qseq", ctypes.c_uint32)]


class mm_mod_t(ctypes.Structure):
    _fields_ = [("code", ctypes.c_char * MM_CODE_LEN), ("context", ctypes.c_char * MM_CODE_LEN),
                ("klass", ctypes.c_uint8 * 256)]


class mm_freq_opts_t(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_int32), ("n_mods", ctypes.c_int32), ("insertions", ctypes.c_int32),
                ("haplotypes", ctypes.c_int32), ("device", ctypes.c_int32), ("n_hp_planes", ctypes.c_int32),
                ("side_capacity", ctypes.c_int64), ("n_wild_planes", ctypes.c_int32), ("view", ctypes.c_int32),
                ("force_fused", ctypes.c_int32), ("view_cap", ctypes.c_int32), ("finalize_by_runs", ctypes.c_int32),
                ("split_bases", ctypes.c_int32), ("coalesce", ctypes.c_int32), ("stream_mode", ctypes.c_int32),
                ("gather_mb", ctypes.c_int32), ("stream_slices", ctypes.c_int32), ("rsvd_opts", ctypes.c_int32), ("mods", mm_mod_t * MM_MAX_MODS)]


class mm_contig_t(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("length", ctypes.c_int64), ("seq", ctypes.c_void_p),
                ("seq_length", ctypes.c_int64)]


class mm_interval_t(ctypes.Structure):
    _fields_ = [("tid", ctypes.c_int32), ("rsvd", ctypes.c_int32), ("begin", ctypes.c_int64),
                ("end", ctypes.c_int64), ("halo", ctypes.c_int64)]


EXPORTS = ["mm_freq_plan_batch", "mm_freq_ticket_batches", "mm_abi_version", "mm_strerror", "mm_freq_create", "mm_freq_submit", "mm_freq_submit_device", "mm_freq_submit_device_now",
           "mm_freq_wait", "mm_freq_host_done", "mm_freq_read_record", "mm_freq_ticket_batch", "mm_view_fetch", "mm_view_fetch_device", "mm_freq_intern_code", "mm_freq_n_codes", "mm_freq_code_name", "mm_freq_finalize", "mm_freq_finalize_device",
           "mm_freq_slab_words", "mm_freq_slab_export", "mm_freq_slab_add", "mm_freq_slab_clear", "mm_freq_slab_export_host", "mm_freq_slab_add_host", "mm_freq_slab_export_ipc", "mm_freq_slab_add_ipc",
           "mm_freq_last_kernel_ms", "mm_freq_stats_enable", "mm_freq_stats_get", "mm_freq_device_bytes", "mm_freq_launch_counts", "mm_freq_reset_counters", "mm_freq_destroy"]

_lib = None


class MinimodHipError(RuntimeError):
    def __init__(self, code, msg, read=None):
        RuntimeError.__init__(self, msg)
        self.code = code
        self.read = read


def _share_hip_runtime_with_torch():
    """PyTorch wheels carry their own libamdhip64 (same soname as /opt/rocm's) and load it by path; a process that has both
    copies only gets a GPU from the one loaded first.  When torch is installed but not imported yet, load its copy first so
    that this library binds to it by soname and a later `import torch` (device buffers for submit_device) still works."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(build=True):
    """Load the HIP library; fails loudly when it is missing (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if build and not os.path.exists(path):
        build_hip()
    if not os.path.exists(path):
        raise MinimodHipError(-1, "HIP extension %s is missing: run __graft_entry__.build()" % path)
    _share_hip_runtime_with_torch()
    L = ctypes.CDLL(path)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.mm_abi_version.restype = i32
    L.mm_strerror.restype = ctypes.c_char_p
    L.mm_strerror.argtypes = [i32]
    L.mm_freq_create.restype = vp
    L.mm_freq_create.argtypes = [ctypes.POINTER(mm_freq_opts_t), i32, ctypes.POINTER(mm_contig_t), i32,
                                 ctypes.POINTER(mm_interval_t), ctypes.c_char_p, ctypes.c_size_t]
    L.mm_freq_submit.restype = i32
    L.mm_freq_submit.argtypes = [vp, ctypes.POINTER(mm_batch_t)]
    L.mm_freq_submit_device.restype = i32
    L.mm_freq_submit_device.argtypes = [vp, ctypes.POINTER(mm_batch_t), vp]
    L.mm_freq_ticket_batches.restype = i32
    L.mm_freq_ticket_batches.argtypes = [vp, i32]
    L.mm_freq_wait.restype = i32
    L.mm_freq_wait.argtypes = [vp, i32, ctypes.POINTER(i32)]
    L.mm_freq_host_done.restype = i32
    L.mm_freq_host_done.argtypes = [vp, i32]
    L.mm_freq_read_record.restype = i32
    L.mm_freq_read_record.argtypes = [vp, i32, i32, vp]
    L.mm_freq_ticket_batch.restype = i32
    L.mm_freq_ticket_batch.argtypes = [vp, i32, ctypes.POINTER(mm_batch_t)]
    for f in ("mm_view_fetch", "mm_view_fetch_device"):
        getattr(L, f).restype = i64
        getattr(L, f).argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(i32)]
    L.mm_freq_plan_batch.restype = i32
    L.mm_freq_plan_batch.argtypes = [vp, i32, vp, i32]
    L.mm_freq_intern_code.restype = i32
    L.mm_freq_intern_code.argtypes = [vp, ctypes.c_char_p]
    L.mm_freq_n_codes.restype = i32
    L.mm_freq_n_codes.argtypes = [vp]
    L.mm_freq_code_name.restype = ctypes.c_char_p
    L.mm_freq_code_name.argtypes = [vp, i32]
    L.mm_freq_finalize.restype = i64
    L.mm_freq_finalize.argtypes = [vp, ctypes.POINTER(vp)]
    L.mm_freq_finalize_device.restype = i64
    L.mm_freq_finalize_device.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.mm_freq_slab_words.restype = i64
    L.mm_freq_slab_words.argtypes = [vp, i64]
    for f in ("mm_freq_slab_export", "mm_freq_slab_add"):
        getattr(L, f).restype = i32
        getattr(L, f).argtypes = [vp, i32, i64, i64, vp, vp]
    L.mm_freq_slab_clear.restype = i32
    L.mm_freq_slab_clear.argtypes = [vp, i32, i64, i64, vp]
    L.mm_freq_last_kernel_ms.restype = ctypes.c_float
    L.mm_freq_last_kernel_ms.argtypes = [vp, i32]
    L.mm_freq_stats_enable.restype = i32
    L.mm_freq_stats_enable.argtypes = [vp, i32]
    L.mm_freq_stats_get.restype = i32
    L.mm_freq_stats_get.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.mm_freq_device_bytes.restype = i64
    L.mm_freq_device_bytes.argtypes = [vp]
    L.mm_freq_launch_counts.restype = i32
    L.mm_freq_launch_counts.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.mm_freq_reset_counters.argtypes = [vp]
    L.mm_freq_destroy.argtypes = [vp]
    if L.mm_abi_version() != MM_ABI_VERSION:
        raise MinimodHipError(-1, "ABI version mismatch")
    _lib = L
    return L


def klass_lut(thresh):
    """The threshold rule of reference src/mod.c:56,1180-1191 for all 256 ML values, in double precision:
    0 ambiguous, 1 called unmodified, 3 called modified."""
    t = float(thresh)
    out = np.zeros(256, dtype=np.uint8)
    for x in range(256):
        p = (x + 0.5) / 256.0
        if p >= t:
            out[x] = 3
        elif p <= 1 - t:
            out[x] = 1
    return out


def plan_batch(reads):
    """Work items for a batch (long reads split into parts, costliest first) as an int32 array."""
    L = load_library()
    reads = np.ascontiguousarray(reads)
    out = np.empty(max(16 * len(reads), 1), dtype=np.int32)
    n = L.mm_freq_plan_batch(reads.ctypes.data, len(reads), out.ctypes.data, len(out))
    if n < 0:
        raise MinimodHipError(-n, "mm_freq_plan_batch failed")
    return out[:n].copy()


def batch_struct(batch, order=None, device=False):
    """numpy batch (dict with reads/cigar/seq/mm/ml) or dict of device pointers -> mm_batch_t."""
    b = mm_batch_t()
    if device:
        for k in ("reads", "cigar", "seq", "mm", "ml"):
            setattr(b, k, int(batch[k]))
        b.order = int(batch.get("order", 0) or 0)
        b.n_order = int(batch.get("n_order", 0) or 0)
        for k in ("n_reads", "n_cigar_words", "n_seq_bytes", "n_mm_bytes", "n_ml_bytes", "max_n_cigar", "max_l_qseq"):
            setattr(b, k, int(batch[k]))
        return b
    rd = batch["reads"]
    b.reads, b.cigar, b.seq = rd.ctypes.data, batch["cigar"].ctypes.data, batch["seq"].ctypes.data
    b.mm, b.ml = batch["mm"].ctypes.data, batch["ml"].ctypes.data
    b.order = order.ctypes.data if order is not None else 0
    b.n_order = len(order) if order is not None else 0
    b.n_reads = len(rd)
    b.n_cigar_words, b.n_seq_bytes = len(batch["cigar"]), len(batch["seq"])
    b.n_mm_bytes, b.n_ml_bytes = len(batch["mm"]), len(batch["ml"])
    b.max_n_cigar = int(rd["n_cigar"].max()) if len(rd) else 0
    b.max_l_qseq = int(rd["l_qseq"].max()) if len(rd) else 0
    return b


class FreqEngine(object):
    """One handle = one `minimod freq` run on one GPU (or, with view=True, one `minimod view` run: per-read rows
    from fetch_view() instead of counters from finalize()).

    mods: [(code, context, threshold)] as parse_mod_codes/parse_mod_threshes of the reference would produce;
    contigs: [(name, target_len, raw_sequence_bytes_or_None)] in BAM-header (tid) order."""

    def __init__(self, mods, contigs, insertions=False, haplotypes=False, device=0, intervals=None,
                 n_hp_planes=0, side_capacity=0, n_wild_planes=0, view=False, force_fused=False, view_cap=0,
                 finalize_by_runs=False, split_bases=0, coalesce=1, stream_mode=0, gather_mb=0, stream_slices=0):
        L = load_library()
        if not (1 <= len(mods) <= MM_MAX_MODS):
            raise MinimodHipError(36, "1..%d modification codes supported" % MM_MAX_MODS)
        o = mm_freq_opts_t()
        o.abi_version, o.n_mods = MM_ABI_VERSION, len(mods)
        o.insertions, o.haplotypes, o.device = int(insertions), int(haplotypes), int(device)
        o.n_hp_planes, o.side_capacity, o.n_wild_planes = int(n_hp_planes), int(side_capacity), int(n_wild_planes)
        o.view = int(view)
        o.force_fused, o.view_cap, o.finalize_by_runs = int(force_fused), int(view_cap), int(finalize_by_runs)
        o.split_bases, o.coalesce, o.gather_mb = int(split_bases), int(coalesce), int(gather_mb)   # (coalesce: 1 = every submit its own launch, this class's default; 0 = the library's 32)
        o.stream_slices = int(stream_slices)
        o.stream_mode = int(stream_mode)   # 0 by launch size, 1 never, 2 always (reads up to split_bases), 3 always + '.' groups from the first launch
        for i, (code, ctx, th) in enumerate(mods):
            o.mods[i].code = code.encode()
            o.mods[i].context = ctx.encode()
            lut = klass_lut(th)
            ctypes.memmove(o.mods[i].klass, lut.ctypes.data, 256)
        self._keep = []
        cs = (mm_contig_t * max(len(contigs), 1))()
        self.names = []
        for t, (name, length, seq) in enumerate(contigs):
            self.names.append(name)
            cs[t].name = name.encode()
            cs[t].length = int(length)
            if seq is not None:
                arr = np.ascontiguousarray(np.frombuffer(seq, dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq)
                self._keep.append(arr)
                cs[t].seq = arr.ctypes.data
                cs[t].seq_length = len(arr)
        ivs, n_iv = None, 0
        if intervals:
            n_iv = len(intervals)
            ivs = (mm_interval_t * n_iv)()
            for i, (tid, b, e, halo) in enumerate(intervals):
                ivs[i].tid, ivs[i].begin, ivs[i].end, ivs[i].halo = int(tid), int(b), int(e), int(halo)
        err = ctypes.create_string_buffer(512)
        self.L = L
        self.h = L.mm_freq_create(ctypes.byref(o), len(contigs), cs, n_iv, ivs, err, 512)
        if not self.h:
            raise MinimodHipError(-1, "mm_freq_create: " + err.value.decode(errors="replace"))
        self._keep = []  # the reference has been uploaded
        self.insertions, self.haplotypes = insertions, haplotypes
        self.wildcard = any(c == "*" for c, _, _ in mods)

    # -- batches
    def intern_codes_from(self, batch):
        """-c '*' only: scan MM group headers on the host so every code string has an index before the kernel."""
        mm = batch["mm"]
        for rd in batch["reads"]:
            s = bytes(mm[int(rd["mm_off"]):int(rd["mm_off"]) + int(rd["mm_len"])])
            for g in s.split(b";"):
                if len(g) < 3:
                    continue
                j = 2
                while j < len(g) and g[j:j + 1] not in (b",", b"?", b"."):
                    j += 1
                codes = g[2:j]
                if not codes:
                    continue
                if codes.isdigit():
                    self.L.mm_freq_intern_code(self.h, codes)
                else:
                    for m in range(len(codes)):
                        self.L.mm_freq_intern_code(self.h, codes[m:])

    def submit(self, batch, order=None):
        if self.wildcard:
            self.intern_codes_from(batch)
        b = batch_struct(batch, order)
        t = self.L.mm_freq_submit(self.h, ctypes.byref(b))
        if t < 0:
            raise MinimodHipError(-t, "mm_freq_submit: " + self.L.mm_strerror(t).decode())
        return t

    def submit_device(self, dev_batch, stream=None):
        # (a caller that submits the same resident windows over and over hands in the struct it made once with batch_struct(..., device=True):
        # building it here costs ~10 us of Python a call, a quarter of a timed step of bench.py)
        b = dev_batch if isinstance(dev_batch, mm_batch_t) else batch_struct(dev_batch, device=True)
        t = self.L.mm_freq_submit_device(self.h, ctypes.byref(b), stream)
        if t < 0:
            raise MinimodHipError(-t, "mm_freq_submit_device: " + self.L.mm_strerror(t).decode())
        return t

    def ticket_batches(self, ticket):
        """Submits gathered into the ticket's launch (mm_freq_opts_t.coalesce)."""
        return int(self.L.mm_freq_ticket_batches(self.h, ticket))

    def wait(self, ticket):
        bad = ctypes.c_int32(-1)
        e = self.L.mm_freq_wait(self.h, ticket, ctypes.byref(bad))
        if e:
            raise MinimodHipError(e, "read %d: %s" % (bad.value, self.L.mm_strerror(e).decode()), bad.value)

    def host_done(self, ticket):
        """The host memory of every batch submitted under the ticket so far may be reused."""
        e = self.L.mm_freq_host_done(self.h, ticket)
        if e:
            raise MinimodHipError(e, "mm_freq_host_done: " + self.L.mm_strerror(e).decode())

    def read_record(self, ticket, index):
        """Read `index` of the ticket's (gathered) batch as the device holds it (a READ_DTYPE scalar)."""
        out = np.zeros(1, dtype=READ_DTYPE)
        e = self.L.mm_freq_read_record(self.h, ticket, int(index), out.ctypes.data)
        if e:
            raise MinimodHipError(e, "mm_freq_read_record: " + self.L.mm_strerror(e).decode())
        return out[0]

    def ticket_batch(self, ticket):
        """the ticket's batch as the device holds it (an mm_batch_t of device pointers)"""
        b = mm_batch_t()
        e = self.L.mm_freq_ticket_batch(self.h, ticket, ctypes.byref(b))
        if e:
            raise MinimodHipError(e, "mm_freq_ticket_batch: " + self.L.mm_strerror(e).decode())
        return b

    def process(self, batch, order=None):
        self.wait(self.submit(batch, order))

    def fetch_view(self, ticket, device=False):
        """Rows of a view-mode ticket in print_view_output order: a VIEW_ROW_DTYPE array, or with device=True
        (device pointer, row count)."""
        bad = ctypes.c_int32(-1)
        p = ctypes.c_void_p()
        f = self.L.mm_view_fetch_device if device else self.L.mm_view_fetch
        n = f(self.h, ticket, ctypes.byref(p), ctypes.byref(bad))
        if n < 0:
            raise MinimodHipError(int(-n), "read %d: %s" % (bad.value, self.L.mm_strerror(int(-n)).decode()), bad.value)
        if device:
            return p.value, int(n)
        if n == 0:
            return np.zeros(0, dtype=VIEW_ROW_DTYPE)
        buf = (ctypes.c_char * (n * VIEW_ROW_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=VIEW_ROW_DTYPE).copy()

    def view(self, batch, order=None):
        return self.fetch_view(self.submit(batch, order))

    def kernel_ms(self, ticket):
        return float(self.L.mm_freq_last_kernel_ms(self.h, ticket))

    def stats_enable(self, on=True):
        self.L.mm_freq_stats_enable(self.h, int(on))

    def stats_get(self):
        out = (ctypes.c_uint64 * 16)()
        r = self.L.mm_freq_stats_get(self.h, out)
        if r:
            raise MinimodHipError(-r, "stats_get failed")
        return {"lookups": int(out[0]), "ml_reads": int(out[1]), "dense_updates": int(out[2]), "side_updates": int(out[3]),
                "stream_done": int(out[4]), "stream_to_tiles": int(out[5]), "stream_to_fused": int(out[6]),
                "phase_cycles": [int(out[i]) for i in range(4, 16)]}

    # -- results
    def finalize(self):
        p = ctypes.c_void_p()
        n = self.L.mm_freq_finalize(self.h, ctypes.byref(p))
        if n < 0:
            raise MinimodHipError(-n, "mm_freq_finalize: " + self.L.mm_strerror(int(n)).decode())
        if n == 0:
            return np.zeros(0, dtype=ROW_DTYPE)
        buf = (ctypes.c_char * (n * ROW_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=ROW_DTYPE).copy()

    def finalize_device(self):
        """mm_freq_finalize_device: (rows in host memory or None, device pointer or None, n) -- exactly one of the first two when n > 0"""
        p, d = ctypes.c_void_p(), ctypes.c_void_p()
        n = self.L.mm_freq_finalize_device(self.h, ctypes.byref(p), ctypes.byref(d))
        if n < 0:
            raise MinimodHipError(-n, "mm_freq_finalize_device: " + self.L.mm_strerror(int(n)).decode())
        if d.value:
            return None, d.value, int(n)
        if n == 0 or not p.value:
            return np.zeros(0, dtype=ROW_DTYPE), None, int(n)
        buf = (ctypes.c_char * (n * ROW_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=ROW_DTYPE).copy(), None, int(n)

    def code_names(self):
        return [self.L.mm_freq_code_name(self.h, i).decode() for i in range(self.L.mm_freq_n_codes(self.h))]

    def reset(self):
        self.L.mm_freq_reset_counters(self.h)

    def launch_counts(self):
        out = (ctypes.c_uint64 * 4)()
        self.L.mm_freq_launch_counts(self.h, out)
        return {"launches": int(out[0]), "stream_launches": int(out[1]), "submits": int(out[2]), "reads": int(out[3])}

    def device_bytes(self):
        return int(self.L.mm_freq_device_bytes(self.h))

    def slab_words(self, length):
        return int(self.L.mm_freq_slab_words(self.h, length))

    def slab_export(self, tid, begin, length, dst_ptr, stream=None):
        r = self.L.mm_freq_slab_export(self.h, tid, begin, length, dst_ptr, stream)
        if r:
            raise MinimodHipError(-r, "slab_export failed")

    def slab_add(self, tid, begin, length, src_ptr, stream=None):
        r = self.L.mm_freq_slab_add(self.h, tid, begin, length, src_ptr, stream)
        if r:
            raise MinimodHipError(-r, "slab_add failed")

    def slab_clear(self, tid, begin, length, stream=None):
        r = self.L.mm_freq_slab_clear(self.h, tid, begin, length, stream)
        if r:
            raise MinimodHipError(-r, "slab_clear failed")

    def close(self):
        if getattr(self, "h", None):
            self.L.mm_freq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
