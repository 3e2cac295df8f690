"""ctypes driver for oracle/freq_oracle.c + Python restatements of the reference's option parsing
and output formatting.  TEST INFRASTRUCTURE ONLY (see oracle/freq_oracle.c header).

Restated here (reference file:line under /root/reference):
  * parse_mod_codes / default contexts / parse_mod_threshes   src/mod.c:99-112,204-398
  * print_freq_header / print_freq_output                    src/mod.c:628-728
"""
import ctypes
import os
import subprocess

import numpy as np

from . import pybam

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(BUILD, "libfreq_oracle.so")

ROW_DTYPE = np.dtype([("tid", "<i4"), ("pos", "<i4"), ("strand", "<i4"), ("code", "<i4"),
                      ("ins_off", "<i4"), ("hp", "<i4"), ("n_called", "<u4"), ("n_mod", "<u4")])

VIEW_DTYPE = np.dtype([("read", "<i8"), ("tid", "<i4"), ("pos", "<i4"), ("strand", "<i4"), ("code", "<i4"),
                       ("ins_off", "<i4"), ("hp", "<i4"), ("read_pos", "<i4"), ("prob", "<u4")])

ERRORS = {1: "hard clip", 2: "unhandled cigar op", 3: "invalid MM base", 4: "invalid MM strand",
          5: "invalid mod code char", 6: "empty mod codes", 7: "mixed mod codes", 8: "skip count too long",
          9: "bad skip count", 10: "read pos out of range", 11: "ML index overrun", 12: "contig not in reference",
          13: "ref pos outside contig", 14: "cigar longer than sequence"}

DEFAULT_CONTEXT = {"*": "*", "m": "CG", "h": "CG", "f": "C", "c": "C", "C": "C", "g": "T", "e": "T", "b": "T",
                   "T": "T", "U": "T", "a": "A", "A": "A", "o": "G", "G": "G", "n": "N", "N": "N"}


def build(force=False):
    src = os.path.join(HERE, "freq_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        os.makedirs(BUILD, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-Wall", "-o", LIB, src, "-lpthread"])
    return LIB


CPU_BIN = os.path.join(BUILD, "freq_cpu")


def build_cpu_cli(force=False):
    """oracle/freq_cpu_main.c: `minimod freq` end to end on the CPU (the oracle behind the product's own readers and
    formatter), for bench.py's cpu_baseline leg.  Links the host library built by minimod_amd.build_all()."""
    root = os.path.dirname(HERE)
    srcs = [os.path.join(HERE, "freq_cpu_main.c"), os.path.join(HERE, "freq_oracle.c")]
    libdir = os.path.join(root, "minimod_amd", "lib")
    deps = srcs + [os.path.join(libdir, "libminimod_host.so")]
    if force or not os.path.exists(CPU_BIN) or any(os.path.getmtime(CPU_BIN) < os.path.getmtime(d) for d in deps):
        os.makedirs(BUILD, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-Wall", "-o", CPU_BIN] + srcs +
                              ["-I", os.path.join(root, "include"), "-I", os.path.join(root, "minimod_amd", "csrc", "host"),
                               "-L", libdir, "-lminimod_host", "-lminimod_hip", "-Wl,-rpath," + libdir, "-lpthread", "-lm", "-lz"])
    return CPU_BIN


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p),
                                 ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_add_contig.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64]
        L.orc_name_contig.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p]
        L.orc_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.orc_n_rows.restype = ctypes.c_int64
        L.orc_n_rows.argtypes = [ctypes.c_void_p]
        L.orc_error_read.restype = ctypes.c_int64
        L.orc_error_read.argtypes = [ctypes.c_void_p]
        L.orc_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_n_codes.argtypes = [ctypes.c_void_p]
        L.orc_code_name.restype = ctypes.c_char_p
        L.orc_code_name.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_set_view.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_n_view_rows.restype = ctypes.c_int64
        L.orc_n_view_rows.argtypes = [ctypes.c_void_p]
        L.orc_view_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code, read):
        Exception.__init__(self, "oracle error %d (%s) at read %d" % (code, ERRORS.get(code, "?"), read))
        self.code = code
        self.read = read


def parse_mod_codes(s):
    """-c string -> [(code, context)] ; restates parse_mod_codes, src/mod.c:204-326."""
    out = []
    i = 0
    if not s:
        s = "m"
    while i < len(s):
        code = ""
        has_alpha = has_num = False
        while i < len(s) and s[i] not in ",[":
            ch = s[i]
            if ch.isalpha() and ch.isascii() or ch == "*":
                has_alpha = True
            elif ch.isdigit():
                has_num = True
            else:
                raise ValueError("Invalid character %s in modification code" % ch)
            code += ch
            i += 1
        if has_alpha and has_num:
            raise ValueError("Modification code %s cannot contain both letters and numbers" % code)
        if i < len(s) and s[i] == "[":
            i += 1
            ctx = ""
            star = False
            while i < len(s) and s[i] != "]":
                ch = s[i]
                if ch == "*":
                    star = True
                elif ch not in "ACGTUNacgtun":
                    raise ValueError("Invalid character %s in context" % ch)
                ch = ch.upper()
                ctx += "T" if ch == "U" else ch
                i += 1
            if i >= len(s):
                raise ValueError("Context not closed with a ]")
            if star and len(ctx) > 1:
                raise ValueError("* should be the only character within [ and ]")
            i += 1
            if i < len(s) and s[i] == ",":
                i += 1
        elif i < len(s) and s[i] == ",":
            ctx = DEFAULT_CONTEXT.get(code, "CG") if len(code) == 1 else "CG"
            i += 1
        else:
            ctx = DEFAULT_CONTEXT.get(code, "CG") if len(code) == 1 else "CG"
        if code in [c for c, _ in out]:
            raise ValueError("Duplicate modification code %s" % code)
        out.append((code, ctx))
    return out


def parse_mod_threshes(s, n_mods):
    """-m string -> [thresh]*n_mods ; restates parse_mod_threshes, src/mod.c:328-398."""
    if not s:
        s = ",".join(["0.8"] * n_mods)
    vals = [float(x) if x else 0.0 for x in s.split(",")]
    for d in vals:
        if d < 0 or d > 1:
            raise ValueError("Modification threshold should be in the range 0.0 to 1.0")
    if len(vals) == 1:
        vals = vals * n_mods
    elif len(vals) != n_mods:
        raise ValueError("Number of modification codes and thresholds do not match")
    return vals


class Oracle(object):
    def __init__(self, mods, thresh, target_names, insertions=False, haplotypes=False):
        """mods: [(code, context)], thresh: [float]; target_names: BAM header contig names by tid."""
        L = lib()
        n = len(mods)
        codes = (ctypes.c_char_p * n)(*[c.encode() for c, _ in mods])
        ctxs = (ctypes.c_char_p * n)(*[c.encode() for _, c in mods])
        th = (ctypes.c_double * n)(*thresh)
        self.names = list(target_names)
        self.h = L.orc_create(n, codes, ctxs, th, int(insertions), int(haplotypes), len(self.names))
        if not self.h:
            raise RuntimeError("orc_create failed")
        for tid, nm in enumerate(self.names):
            L.orc_name_contig(self.h, tid, nm.encode())
        self.insertions, self.haplotypes = insertions, haplotypes

    def add_contig(self, name, raw):
        raw = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw)
        tid = self.names.index(name)
        lib().orc_add_contig(self.h, tid, name.encode(), raw.ctypes.data, len(raw))

    def process(self, batch, threads=1):
        rd = batch["reads"]
        e = lib().orc_process(self.h, rd.ctypes.data, len(rd), batch["cigar"].ctypes.data, batch["seq"].ctypes.data,
                              batch["mm"].ctypes.data, batch["ml"].ctypes.data, threads)
        if e:
            raise OracleError(e, lib().orc_error_read(self.h))

    def rows(self):
        n = lib().orc_n_rows(self.h)
        out = np.zeros(n, dtype=ROW_DTYPE)
        if n:
            lib().orc_rows(self.h, out.ctypes.data)
        return out

    def set_view(self, on=True):
        lib().orc_set_view(self.h, int(on))

    def view_rows(self):
        n = lib().orc_n_view_rows(self.h)
        out = np.zeros(n, dtype=VIEW_DTYPE)
        if n:
            lib().orc_view_rows(self.h, out.ctypes.data)
        return out

    def code_names(self):
        return [lib().orc_code_name(self.h, i).decode() for i in range(lib().orc_n_codes(self.h))]

    def close(self):
        if self.h:
            lib().orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _f6(x):
    return "%f" % x


def format_rows(rows, names, code_names, bedmethyl=False, insertions=False, haplotypes=False, header=True):
    """print_freq_header + print_freq_output, src/mod.c:628-728.  Returns the output text."""
    out = []
    if not bedmethyl and header:
        h = "contig\tstart\tend\tstrand\tn_called\tn_mod\tfreq\tmod_code"
        if insertions:
            h += "\tins_offset"
        if haplotypes:
            h += "\thaplotype"
        out.append(h + "\n")
    for r in rows:
        contig = names[r["tid"]]
        pos = int(r["pos"])
        strand = "-" if r["strand"] else "+"
        nc, nm = int(r["n_called"]), int(r["n_mod"])
        code = code_names[r["code"]]
        if bedmethyl:
            f = float(nm) * 100 / nc
            out.append("%s\t%d\t%d\t%s\t%d\t%s\t%d\t%d\t255,0,0\t%d\t%s\n" %
                       (contig, pos, pos + 1, code, nc, strand, pos, pos + 1, nc, _f6(f)))
        else:
            f = float(nm) / nc
            line = "%s\t%d\t%d\t%s\t%d\t%d\t%s\t%s" % (contig, pos, pos, strand, nc, nm, _f6(f), code)
            if insertions:
                line += "\t%d" % int(r["ins_off"])
            if haplotypes:
                line += "\t*" if r["hp"] < 0 else "\t%d" % int(r["hp"])
            out.append(line + "\n")
    return "".join(out)


def format_view(rows, qnames, names, code_names, insertions=False, haplotypes=False, header=True):
    """print_view_header + print_view_output, src/mod.c:545-626.  qnames: read names by running read index."""
    out = []
    if header:
        h = "ref_contig\tref_pos\tstrand\tread_id\tread_pos\tmod_code\tmod_prob"
        if insertions:
            h += "\tins_offset"
        if haplotypes:
            h += "\thaplotype"
        out.append(h + "\n")
    for r in rows:
        q = qnames[int(r["read"])]
        q = q.decode() if isinstance(q, bytes) else q
        line = "%s\t%d\t%s\t%s\t%d\t%s\t%s" % (names[r["tid"]], int(r["pos"]), "-" if r["strand"] else "+", q,
                                                int(r["read_pos"]), code_names[r["code"]],
                                                _f6((int(r["prob"]) + 0.5) / 256.0))
        if insertions:
            line += "\t%d" % int(r["ins_off"])
        if haplotypes:
            line += "\t%d" % int(r["hp"])
        out.append(line + "\n")
    return "".join(out)


def view(bam_path, contigs, c="m", insertions=False, haplotypes=False, allow_secondary=False,
         skip_supplementary=False, threads=1, K=512, B=20 * 1000 * 1000):
    """End-to-end `minimod view` on the oracle: returns (rows, qnames, names, code_names)."""
    mods = parse_mod_codes(c)
    th = parse_mod_threshes(None, len(mods))
    orc = None
    qnames = []
    for bam, batch, _st in pybam.load_batches(bam_path, K=K, B=B, allow_secondary=allow_secondary,
                                             skip_supplementary=skip_supplementary):
        if orc is None:
            orc = Oracle(mods, th, bam.target_name, insertions, haplotypes)
            orc.set_view(True)
            for name, seq in contigs.items():
                if name in bam.target_name:
                    orc.add_contig(name, seq)
        if len(batch["reads"]):
            orc.process(batch, threads)
            qnames += batch["qnames"]
    rows = orc.view_rows()
    names, codes = orc.names, orc.code_names()
    orc.close()
    return rows, qnames, names, codes


def pseudo_reference(npz_path):
    """Expand a (pos, base) patch list from tests/golden into a full contig (all-N elsewhere)."""
    z = np.load(npz_path)
    seq = np.full(int(z["length"]), ord("N"), dtype=np.uint8)
    seq[z["pos"]] = z["base"]
    return str(z["contig"]), seq


def freq(bam_path, contigs, c="m", m=None, insertions=False, haplotypes=False, allow_secondary=False,
         skip_supplementary=False, threads=1, K=512, B=20 * 1000 * 1000):
    """End-to-end `minimod freq` on the oracle: returns (rows, names, code_names)."""
    mods = parse_mod_codes(c)
    th = parse_mod_threshes(m, len(mods))
    orc = None
    for bam, batch, _st in pybam.load_batches(bam_path, K=K, B=B, allow_secondary=allow_secondary,
                                             skip_supplementary=skip_supplementary):
        if orc is None:
            orc = Oracle(mods, th, bam.target_name, insertions, haplotypes)
            for name, seq in contigs.items():
                if name in bam.target_name:
                    orc.add_contig(name, seq)
        if len(batch["reads"]):
            orc.process(batch, threads)
    rows = orc.rows()
    names, codes = orc.names, orc.code_names()
    orc.close()
    return rows, names, codes
