#!/bin/bash
# A variant of the device library that differs in ONE translation unit's build flags: tools/var_unit.sh <unit: bgzf_api | ingest_api | tie_api> <name> <-D flags...>
# -> minimod_amd/lib/var/<name>.so (run with MM_HIP_LIB=minimod_amd/lib/var/<name>.so)
cd "$(dirname "$0")/.." || exit 1
unit=$1; name=$2; shift; shift
mkdir -p minimod_amd/lib/var
h=$(python -c "from minimod_amd import build as B; print(B.library_source_hash())")
extra=""; [ $unit = bgzf_api ] && extra="-DMM_SOURCE_HASH=\"$h\""
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c -I include -o minimod_amd/lib/obj/$unit.var_$name.o minimod_amd/csrc/$unit.hip $extra "$@" || exit 1
objs=""
for u in freq_api_k0 freq_api_k1 freq_api_k2 freq_dispatch bgzf_api ingest_api tie_api; do
  if [ $u = $unit ]; then objs="$objs minimod_amd/lib/obj/$unit.var_$name.o"; else objs="$objs minimod_amd/lib/obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o minimod_amd/lib/var/$name.so $objs && echo built minimod_amd/lib/var/$name.so
