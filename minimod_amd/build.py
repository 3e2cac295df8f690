"""Build the in-tree native libraries (hipcc for the gfx950 device library, gcc for the C host library)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
INCLUDE = os.path.join(ROOT, "include")


def lib_path(name="libminimod_hip.so"):
    if name == "libminimod_hip.so" and os.environ.get("MM_HIP_LIB"):   # developer switch: another build of the device library
        return os.environ["MM_HIP_LIB"]
    return os.path.join(LIBDIR, name)


def source_hash():
    """sha256 over the sources of the device library's freq / view path (what a measurement made with rocprofv3 outside bench.py
    is stamped with; the BGZF inflate is another translation unit and none of those launches)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(f for f in glob.glob(os.path.join(CSRC, "*.hip*")) if not os.path.basename(f).startswith(("bgzf_", "ingest_", "tie_", "fmt_", "summary_"))) + [os.path.join(INCLUDE, "minimod_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _hash_files(files):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def library_source_hash():
    """sha256 over EVERY source of the device library (csrc/*.hip*, include/*.h): what the built library carries
    (mm_build_source_hash) and what build() / smoke() compare it with -- a shipped .so made from other sources is rebuilt, not reused."""
    import glob
    return _hash_files(sorted(glob.glob(os.path.join(CSRC, "*.hip*")) + glob.glob(os.path.join(CSRC, "fmt_core.h")) + glob.glob(os.path.join(CSRC, "devmem.*")) + glob.glob(os.path.join(INCLUDE, "*.h"))))


def built_library_hash(path=None):
    """the hash the library at `path` was built from (None: it does not say).  Read from the file's bytes -- the marker string
    'MMSRCHASH=<hex>' the BGZF unit carries -- not through dlopen: glibc caches a loaded library by path name, so a later CDLL of the
    relinked file would hand back the old mapping."""
    import re
    try:
        with open(path or lib_path(), "rb") as f:
            m = re.search(rb"MMSRCHASH=([0-9a-f]{16})", f.read())
        return m.group(1).decode() if m else None
    except OSError:
        return None


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


LAST_BUILD = {"compiled": [], "reused": [], "linked": False}   # what the last build_hip() did (build() prints it: a record of whether anything was compiled)


def build_hip(force=False, verbose=False, always=()):
    """hipcc cross-compiles for gfx950 without a GPU present.  Two translation units, compiled to objects of their own (the
    freq path takes 80 s, the BGZF inflate 6 s) and linked into the one library."""
    out = lib_path()
    freq_srcs = [os.path.join(CSRC, "freq_api.hip"), os.path.join(CSRC, "freq_kernels.hip.h"), os.path.join(CSRC, "freq_tiles.hip.h"), os.path.join(CSRC, "freq_stream.hip.h"), os.path.join(CSRC, "view_kernels.hip.h"),
                 os.path.join(CSRC, "sort_kernels.hip.h"),
                 os.path.join(INCLUDE, "minimod_hip.h")]
    bgzf_srcs = [os.path.join(CSRC, "bgzf_api.hip"), os.path.join(CSRC, "bgzf_kernels.hip.h"), os.path.join(INCLUDE, "minimod_bgzf.h")]
    ingest_srcs = [os.path.join(CSRC, "ingest_api.hip"), os.path.join(CSRC, "ingest_kernels.hip.h"), os.path.join(INCLUDE, "minimod_ingest.h"),
                   os.path.join(INCLUDE, "minimod_bgzf.h"), os.path.join(INCLUDE, "minimod_hip.h")]
    tie_srcs = [os.path.join(CSRC, "tie_api.hip"), os.path.join(CSRC, "tie_kernels.hip.h"), os.path.join(INCLUDE, "minimod_tie.h"), os.path.join(INCLUDE, "minimod_hip.h")]
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    defs = os.environ.get("MM_HIP_DEFS", "").split()   # build-time experiments: -DMM_TILE_CHARS=... etc.
    objs = []
    relink = force or not os.path.exists(out)
    full = library_source_hash()
    freq_srcs = freq_srcs + [os.path.join(CSRC, "freq_kinds.h")]
    dispatch_srcs = [os.path.join(CSRC, "freq_dispatch.cpp"), os.path.join(INCLUDE, "minimod_hip.h")]
    # (translation unit, its sources, extra flags): the freq / view path once per reference-word kind (csrc/freq_kinds.h), the public
    # names that forward to them, the BGZF inflate, the ingestion
    # (the freq path without machine-level loop-invariant code motion and with sinking to avoid spills: k_stream_reads keeps more
    # wave-uniform state than there are scalar registers, and what LICM hoists out of its loops is parked in vector-register lanes and
    # scratch -- with these two flags the hot instantiation has 0 bytes of scratch instead of 68 and 18 % fewer v_readlane; C2 35.1 ->
    # 33.9 us per batch, C3 -2.3 %, C5 -2.2 %, tools/ab.sh)
    freq_flags = ["-mllvm", "-disable-machine-licm", "-mllvm", "-sink-insts-to-avoid-spills"]
    devmem = os.path.join(CSRC, "devmem.h")   # (the block-keeping allocator every unit allocates through)
    freq_srcs, bgzf_srcs, ingest_srcs, tie_srcs = freq_srcs + [devmem], bgzf_srcs + [devmem], ingest_srcs + [devmem], tie_srcs + [devmem]
    units = [("freq_api_k%d" % k, freq_srcs, ["-DMM_KIND=%d" % k] + freq_flags) for k in (0, 1, 2)] + \
            [("freq_dispatch", dispatch_srcs, []), ("devmem", [os.path.join(CSRC, "devmem.cpp"), devmem], []), ("bgzf_api", bgzf_srcs, ['-DMM_SOURCE_HASH="%s"' % full]), ("ingest_api", ingest_srcs, []), ("tie_api", tie_srcs + [os.path.join(CSRC, "fmt_api.hip.h"), os.path.join(CSRC, "fmt_core.h"), os.path.join(CSRC, "summary_api.hip.h"), os.path.join(INCLUDE, "minimod_summary.h")], [])]
    todo = []
    LAST_BUILD["compiled"], LAST_BUILD["reused"], LAST_BUILD["linked"] = [], [], False
    for name, srcs, extra in units:
        obj = os.path.join(objdir, name + (".%s.o" % "_".join(defs).replace("-D", "").replace("=", "") if defs else ".o"))
        # an object is reused only when it was made from these very bytes and flags (their hash is kept beside it): time stamps say
        # nothing after a checkout or a copy to another machine
        want = _hash_files(srcs) + "+" + "+".join(extra + defs)
        side = obj + ".srchash"
        have = open(side).read().strip() if os.path.exists(side) and os.path.exists(obj) else None
        if force or have != want or name in always:
            cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-c", "-I", INCLUDE, "-o", obj, srcs[0]] + extra + defs
            todo.append((cmd, side, want))
            LAST_BUILD["compiled"].append(name)
        else:
            LAST_BUILD["reused"].append(name)
        objs.append(obj)
    # the three copies of the freq path take half a minute each: side by side
    procs = []
    for cmd, side, want in todo:
        if verbose:
            print(" ".join(cmd))
        procs.append((subprocess.Popen(cmd), cmd, side, want))
    for pr, cmd, side, want in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
        with open(side, "w") as f:
            f.write(want)
        relink = True
    if relink or _stale(out, objs) or built_library_hash(out) != full:
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        LAST_BUILD["linked"] = True
    return out


def build_host(force=False, verbose=False):
    hostdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hostdir):
        return None
    mk = os.path.join(hostdir, "Makefile")
    if os.path.exists(mk):
        subprocess.check_call(["make", "-s", "-C", hostdir] + (["-B"] if force else []))
    return hostdir


def build_all(force=False, verbose=False, always=()):
    build_hip(force, verbose, always)
    build_host(force, verbose)
