#!/bin/bash
# A variant of the device library that differs in the BGZF inflate's build flags only: tools/var_bgzf.sh <name> "<-D flags>"
# -> minimod_amd/lib/var/<name>.so (run with MM_HIP_LIB=minimod_amd/lib/var/<name>.so)
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p minimod_amd/lib/var
h=$(python -c "from minimod_amd import build as B; print(B.library_source_hash())")
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c -I include -o minimod_amd/lib/obj/bgzf_api.var_$name.o minimod_amd/csrc/bgzf_api.hip -DMM_SOURCE_HASH="\"$h\"" $* || exit 1
objs=$(ls minimod_amd/lib/obj/freq_api_k?.o minimod_amd/lib/obj/freq_dispatch.o minimod_amd/lib/obj/ingest_api.o minimod_amd/lib/obj/tie_api.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o minimod_amd/lib/var/$name.so $objs minimod_amd/lib/obj/bgzf_api.var_$name.o && echo built minimod_amd/lib/var/$name.so
