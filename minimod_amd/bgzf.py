"""ctypes mirror of include/minimod_bgzf.h (the device-side BGZF inflate) for the tests and the bench: plumbing, no logic."""
import ctypes
import struct

import numpy as np

from . import build

BLOCK_DTYPE = np.dtype([("c_off", "<u4"), ("c_len", "<u4"), ("o_off", "<u4"), ("isize", "<u4"), ("crc", "<u4")])
EXPORTS = ["mm_bgzf_create", "mm_bgzf_destroy", "mm_bgzf_host_alloc", "mm_bgzf_host_free", "mm_bgzf_staging", "mm_bgzf_blocks",
           "mm_bgzf_submit", "mm_bgzf_wait", "mm_bgzf_times", "mm_bgzf_inflate_device"]   # (+ mm_build_source_hash: not an mm_bgzf_ name)
_L = None


def lib():
    global _L
    if _L is None:
        L = ctypes.CDLL(build.lib_path())
        L.mm_bgzf_create.restype = ctypes.c_void_p
        L.mm_bgzf_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        L.mm_bgzf_destroy.argtypes = [ctypes.c_void_p]
        L.mm_bgzf_host_alloc.restype = ctypes.c_void_p
        L.mm_bgzf_host_alloc.argtypes = [ctypes.c_size_t]
        L.mm_bgzf_host_free.argtypes = [ctypes.c_void_p]
        L.mm_bgzf_staging.restype = ctypes.c_void_p
        L.mm_bgzf_staging.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        L.mm_bgzf_blocks.restype = ctypes.c_void_p
        L.mm_bgzf_blocks.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        L.mm_bgzf_submit.restype = ctypes.c_int32
        L.mm_bgzf_submit.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
        L.mm_bgzf_wait.restype = ctypes.c_int32
        L.mm_bgzf_wait.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.POINTER(ctypes.c_int32))]
        L.mm_bgzf_times.restype = ctypes.c_int32
        L.mm_bgzf_times.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_float)]
        _L = L
    return _L


def split_bgzf(data):
    """The BGZF blocks of a file's bytes: [(payload, isize, crc)] (RFC 1952 member with the BC subfield, SAM spec 4.1)."""
    out, pos, n = [], 0, len(data)
    while pos < n:
        if n - pos < 18 or data[pos] != 31 or data[pos + 1] != 139 or data[pos + 2] != 8 or not (data[pos + 3] & 4):
            raise ValueError("not a BGZF block at %d" % pos)
        xlen = struct.unpack_from("<H", data, pos + 10)[0]
        x, xe, bsize = pos + 12, pos + 12 + xlen, None
        while x + 4 <= xe:
            sl = struct.unpack_from("<H", data, x + 2)[0]
            if data[x] == 66 and data[x + 1] == 67 and sl == 2:
                bsize = struct.unpack_from("<H", data, x + 4)[0]
            x += 4 + sl
        if bsize is None:
            raise ValueError("no BC subfield at %d" % pos)
        total = bsize + 1
        crc, isize = struct.unpack_from("<II", data, pos + total - 8)
        out.append((bytes(data[pos + 12 + xlen:pos + total - 8]), isize, crc))
        pos += total
    return out


class Inflater:
    """One handle, `slots` launches in flight; inflate() is the synchronous convenience the tests use."""

    def __init__(self, device=0, slots=2, max_blocks=4096, max_cbytes=96 << 20, max_obytes=256 << 20):
        err = ctypes.create_string_buffer(256)
        self.h = lib().mm_bgzf_create(device, slots, max_blocks, max_cbytes, max_obytes, err, 256)
        if not self.h:
            raise RuntimeError("mm_bgzf_create: " + err.value.decode())
        self.slots, self.max_blocks, self.max_cbytes, self.max_obytes = slots, max_blocks, max_cbytes, max_obytes
        self._out = {}

    def close(self):
        if self.h:
            for p in self._out.values():
                lib().mm_bgzf_host_free(p)
            lib().mm_bgzf_destroy(self.h)
            self.h = None

    def staging(self, slot):
        return np.ctypeslib.as_array(ctypes.cast(lib().mm_bgzf_staging(self.h, slot), ctypes.POINTER(ctypes.c_uint8)), shape=(self.max_cbytes + 64,))

    def blocks(self, slot):
        raw = np.ctypeslib.as_array(ctypes.cast(lib().mm_bgzf_blocks(self.h, slot), ctypes.POINTER(ctypes.c_uint8)), shape=(self.max_blocks * BLOCK_DTYPE.itemsize,))
        return raw.view(BLOCK_DTYPE)

    def out_buffer(self, slot):
        if slot not in self._out:
            p = lib().mm_bgzf_host_alloc(self.max_obytes + 64)
            if not p:
                raise MemoryError("pinned allocation failed")
            self._out[slot] = p
        return np.ctypeslib.as_array(ctypes.cast(self._out[slot], ctypes.POINTER(ctypes.c_uint8)), shape=(self.max_obytes + 64,))

    def fill(self, slot, blocks):
        """blocks: [(payload, isize, crc)] -> (n, cbytes, obytes) written into the slot's staging and block records"""
        st, br = self.staging(slot), self.blocks(slot)
        c = o = 0
        for i, (payload, isize, crc) in enumerate(blocks):
            st[c:c + len(payload)] = np.frombuffer(payload, dtype=np.uint8)
            br[i] = (c, len(payload), o, isize, crc)
            c += len(payload)
            o += isize
        return len(blocks), c, o

    def submit(self, slot, n, cbytes, obytes):
        out = self.out_buffer(slot)
        r = lib().mm_bgzf_submit(self.h, slot, n, cbytes, obytes, out.ctypes.data)
        if r:
            raise RuntimeError("mm_bgzf_submit: %d" % r)

    def wait(self, slot, n):
        st = ctypes.POINTER(ctypes.c_int32)()
        r = lib().mm_bgzf_wait(self.h, slot, ctypes.byref(st))
        if r:
            raise RuntimeError("mm_bgzf_wait: %d" % r)
        return np.ctypeslib.as_array(st, shape=(max(n, 1),))[:n].copy()

    def times(self, slot):
        ms = (ctypes.c_float * 4)()
        r = lib().mm_bgzf_times(self.h, slot, ms)
        return None if r else {"h2d_ms": ms[0], "inflate_ms": ms[1], "crc_ms": ms[2], "d2h_ms": ms[3]}

    def inflate(self, blocks, slot=0):
        """[(payload, isize, crc)] -> (list of decoded bytes per block, status array)"""
        n, c, o = self.fill(slot, blocks)
        self.submit(slot, n, c, o)
        status = self.wait(slot, n)
        out, br = self.out_buffer(slot), self.blocks(slot)
        return [bytes(out[int(br[i]["o_off"]):int(br[i]["o_off"]) + int(br[i]["isize"])]) for i in range(n)], status
