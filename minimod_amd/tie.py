"""ctypes mirror of include/minimod_tie.h: the device-side replay of the order in which the reference prints rows that tie on
(contig, start).  Bindings for tests and bench.py; the CLI links the library directly (csrc/host/freq_main.c)."""
import ctypes

import numpy as np

from . import engine as E


class mm_tie_opts_t(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_int32), ("device", ctypes.c_int32), ("insertions", ctypes.c_int32), ("haplotypes", ctypes.c_int32),
                ("n_contigs", ctypes.c_int32), ("rsvd", ctypes.c_int32)]


class mm_fmt_opts_t(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_int32), ("device", ctypes.c_int32), ("bedmethyl", ctypes.c_int32), ("insertions", ctypes.c_int32),
                ("haplotypes", ctypes.c_int32), ("n_contigs", ctypes.c_int32), ("n_codes", ctypes.c_int32), ("rsvd", ctypes.c_int32)]


MM_TIE_ABI_VERSION = 1
_bound = [False]


def _lib():
    L = E.load_library()
    if not _bound[0]:
        L.mm_tie_create.restype = ctypes.c_void_p
        L.mm_tie_create.argtypes = [ctypes.POINTER(mm_tie_opts_t), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        L.mm_tie_set_codes.restype = ctypes.c_int32
        L.mm_tie_set_codes.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.mm_tie_add_launch.restype = ctypes.c_int32
        L.mm_tie_add_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.mm_tie_order_rows.restype = ctypes.c_int32
        L.mm_tie_order_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.mm_tie_order_rows2.restype = ctypes.c_int32
        L.mm_tie_order_rows2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        L.mm_tie_sequence_size.restype = ctypes.c_int64
        L.mm_tie_sequence_size.argtypes = [ctypes.c_void_p]
        L.mm_tie_sequence.restype = ctypes.c_int64
        L.mm_tie_sequence.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.mm_tie_failed.restype = ctypes.c_uint32
        L.mm_tie_failed.argtypes = [ctypes.c_void_p]
        L.mm_tie_device_bytes.restype = ctypes.c_int64
        L.mm_tie_device_bytes.argtypes = [ctypes.c_void_p]
        L.mm_tie_destroy.argtypes = [ctypes.c_void_p]
        L.mm_tie_order_plain.restype = ctypes.c_int32
        L.mm_tie_order_plain.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.mm_tie_last_stats.argtypes = [ctypes.c_void_p]
        L.mm_fmt_create.restype = ctypes.c_void_p
        L.mm_fmt_create.argtypes = [ctypes.POINTER(mm_fmt_opts_t), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        L.mm_fmt_rows.restype = ctypes.c_int64
        L.mm_fmt_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]
        L.mm_fmt_rows_device.restype = ctypes.c_int64
        L.mm_fmt_rows_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]
        L.mm_fmt_last_kernel_ms.restype = ctypes.c_float
        L.mm_fmt_last_kernel_ms.argtypes = [ctypes.c_void_p]
        L.mm_fmt_destroy.argtypes = [ctypes.c_void_p]
        _bound[0] = True
    return L


class TieReplay(object):
    """One run's replay: add_launch() per launch of a view=2 handle, order_rows() at the end."""

    def __init__(self, names, lengths, insertions=False, haplotypes=False, device=0):
        self.L = _lib()
        o = mm_tie_opts_t(MM_TIE_ABI_VERSION, int(device), int(insertions), int(haplotypes), len(names), 0)
        self._names = (ctypes.c_char_p * max(1, len(names)))(*[n.encode() if isinstance(n, str) else n for n in names])
        self._lens = (ctypes.c_int64 * max(1, len(names)))(*[int(x) for x in lengths])
        err = ctypes.create_string_buffer(512)
        self.h = self.L.mm_tie_create(ctypes.byref(o), self._names, self._lens, err, 512)
        if not self.h:
            raise RuntimeError("mm_tie_create: " + err.value.decode())

    def set_codes(self, codes, luts):
        """codes: code strings by index; luts: per code the 256-entry uint8 class table of the mod it counts for"""
        cn = (ctypes.c_char_p * max(1, len(codes)))(*[c.encode() for c in codes])
        self._luts = [np.ascontiguousarray(l, np.uint8) for l in luts]
        kl = (ctypes.c_void_p * max(1, len(codes)))(*[l.ctypes.data for l in self._luts])
        r = self.L.mm_tie_set_codes(self.h, len(codes), cn, kl)
        if r:
            raise RuntimeError("mm_tie_set_codes: %d" % r)

    def add_launch(self, dev_batch_struct, rows_ptr, n_rows):
        return int(self.L.mm_tie_add_launch(self.h, ctypes.byref(dev_batch_struct), rows_ptr, int(n_rows), None))

    def order_rows(self, rows):
        """rows: ROW_DTYPE array (mm_freq_finalize's) -> permutation (uint32) putting them into the reference's printing order, or None"""
        rows = np.ascontiguousarray(rows)
        perm = np.zeros(max(1, len(rows)), np.uint32)
        r = self.L.mm_tie_order_rows(self.h, rows.ctypes.data, len(rows), perm.ctypes.data)
        self.last_rc = int(r)
        return perm[:len(rows)] if r == 0 else None

    def sequence(self):
        """every key stamped so far in first-insertion order: (keys as ROW_DTYPE rows without counts, hashes, put_after_last) or None"""
        n = int(self.L.mm_tie_sequence_size(self.h))
        if n < 0:
            return None
        keys = np.zeros(max(1, n), E.ROW_DTYPE)
        hsh = np.zeros(max(1, n), np.uint32)
        pal = ctypes.c_int32(0)
        r = self.L.mm_tie_sequence(self.h, keys.ctypes.data, hsh.ctypes.data, n, ctypes.byref(pal))
        return (keys[:n], hsh[:n], int(pal.value)) if r == n else None

    def failed(self):
        return int(self.L.mm_tie_failed(self.h))

    def stats(self):
        st = np.zeros(8, np.uint64)
        self.L.mm_tie_last_stats(st.ctypes.data)
        return st

    def close(self):
        if self.h:
            self.L.mm_tie_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RowFormatter(object):
    """print_freq_output's row text made on the device (mm_fmt_rows)."""

    def __init__(self, names, codes, bedmethyl=False, insertions=False, haplotypes=False, device=0):
        self.L = _lib()
        o = mm_fmt_opts_t(MM_TIE_ABI_VERSION, int(device), int(bedmethyl), int(insertions), int(haplotypes), len(names), len(codes), 0)
        cn = (ctypes.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        cc = (ctypes.c_char_p * max(1, len(codes)))(*[c.encode() for c in codes])
        err = ctypes.create_string_buffer(512)
        self.h = self.L.mm_fmt_create(ctypes.byref(o), cn, cc, err, 512)
        if not self.h:
            raise RuntimeError("mm_fmt_create: " + err.value.decode())

    def format(self, rows):
        rows = np.ascontiguousarray(rows)
        p = ctypes.c_void_p()
        n = self.L.mm_fmt_rows(self.h, rows.ctypes.data, len(rows), ctypes.byref(p))
        if n < 0:
            raise RuntimeError("mm_fmt_rows: %d" % n)
        return ctypes.string_at(p.value, n) if n else b""

    def format_device(self, device_ptr, n):
        """rows that are in GPU memory already (mm_freq_finalize_device's): mm_fmt_rows_device"""
        p = ctypes.c_void_p()
        k = self.L.mm_fmt_rows_device(self.h, ctypes.c_void_p(device_ptr), int(n), ctypes.byref(p))
        if k < 0:
            raise RuntimeError("mm_fmt_rows_device: %d" % k)
        return ctypes.string_at(p.value, k) if k else b""

    def kernel_ms(self):
        return float(self.L.mm_fmt_last_kernel_ms(self.h))

    def close(self):
        if self.h:
            self.L.mm_fmt_destroy(self.h)
            self.h = None
