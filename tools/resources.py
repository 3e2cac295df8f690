#!/usr/bin/env python3
"""Register, scratch and LDS use of every kernel in the built device library's freq objects, from the code objects' metadata
(what DESIGN's kernel table quotes): tools/resources.py [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    """[(name, {field: int})] of one host object with an embedded gfx950 code object"""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co]).decode()
    out = []
    for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        f = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in
             ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "group_segment_fixed_size")}
        out.append((name, f))
    return out


def demangled(name):
    try:
        return subprocess.check_output(["c++filt", name]).decode().strip().split("(")[0]
    except (OSError, subprocess.CalledProcessError):
        return name


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else "k_stream_reads"
    for k in (0, 1, 2):
        obj = os.path.join(ROOT, "minimod_amd", "lib", "obj", "freq_api_k%d.o" % k)
        if not os.path.exists(obj):
            continue
        for name, f in kernels(obj):
            if flt in name:
                print("%-78s vgpr %3d sgpr %3d scratch %4d B spilled sgpr %3d vgpr %3d lds %6d" % (
                    demangled(name)[-78:], f["vgpr_count"], f["sgpr_count"], f["private_segment_fixed_size"],
                    f["sgpr_spill_count"], f["vgpr_spill_count"], f["group_segment_fixed_size"]))
