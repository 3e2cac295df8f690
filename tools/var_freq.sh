#!/bin/bash
# A variant of the device library whose RefNib copy of the freq path (kind 0: the C2 / C5 / view instantiations) is compiled with other flags:
# tools/var_freq.sh <name> <flags...>  -> minimod_amd/lib/var/<name>.so   (MM_HIP_LIB=...; tools/ab.sh 3 "" base <name>)
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p minimod_amd/lib/var
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c -I include -o minimod_amd/lib/obj/freq_api_k0.var_$name.o minimod_amd/csrc/freq_api.hip -DMM_KIND=0 "$@" || exit 1
objs="minimod_amd/lib/obj/freq_api_k0.var_$name.o minimod_amd/lib/obj/freq_api_k1.o minimod_amd/lib/obj/freq_api_k2.o minimod_amd/lib/obj/freq_dispatch.o minimod_amd/lib/obj/devmem.o minimod_amd/lib/obj/bgzf_api.o minimod_amd/lib/obj/ingest_api.o minimod_amd/lib/obj/tie_api.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o minimod_amd/lib/var/$name.so $objs && echo built minimod_amd/lib/var/$name.so
