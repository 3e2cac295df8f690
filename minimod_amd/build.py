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
    """sha256 over the device library's sources (what a measurement made with rocprofv3 outside bench.py is stamped with)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip*"))) + [os.path.join(INCLUDE, "minimod_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU present."""
    out = lib_path()
    srcs = [os.path.join(CSRC, "freq_api.hip"), os.path.join(CSRC, "freq_kernels.hip.h"), os.path.join(CSRC, "freq_tiles.hip.h"), os.path.join(CSRC, "freq_stream.hip.h"), os.path.join(CSRC, "view_kernels.hip.h"),
            os.path.join(CSRC, "sort_kernels.hip.h"),
            os.path.join(INCLUDE, "minimod_hip.h")]
    if force or _stale(out, srcs):
        os.makedirs(LIBDIR, exist_ok=True)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-I", INCLUDE,
               "-o", out, srcs[0]] + os.environ.get("MM_HIP_DEFS", "").split()   # build-time experiments: -DMM_TILE_CHARS=... etc.
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_host(force=False, verbose=False):
    hostdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hostdir):
        return None
    mk = os.path.join(hostdir, "Makefile")
    if os.path.exists(mk):
        subprocess.check_call(["make", "-s", "-C", hostdir] + (["-B"] if force else []))
    return hostdir


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_host(force, verbose)
