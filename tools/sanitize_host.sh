#!/bin/bash
# Build the C ingestion path (bamio.c + loader.c) with AddressSanitizer+UBSan and with ThreadSanitizer and run it over the
# bundled BAMs (and any BAMs given as arguments).  CPU only: GPU sanitizers are not available on this pool.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
H="$here/../minimod_amd/csrc/host"
out="${TMPDIR:-/tmp}/mm_sanitize"
mkdir -p "$out"
for mode in address,undefined thread; do
    bin="$out/loader_bench_${mode%%,*}"
    gcc -O1 -g -fsanitize=$mode -fno-omit-frame-pointer -std=gnu99 -I"$H" -I"$here/../include" -o "$bin" "$here/loader_bench.c" "$H/loader.c" "$H/bamio.c" -lz -lpthread
    for f in "$here"/../tests/golden/data/*.bam "$@"; do
        ASAN_OPTIONS=detect_leaks=1 "$bin" "$f" 5 > "$out/last.log" 2>&1 || { cat "$out/last.log"; echo "FAILED ($mode): $f"; exit 1; }
        if grep -q "ERROR: \|WARNING: ThreadSanitizer\|runtime error" "$out/last.log"; then cat "$out/last.log"; echo "REPORT ($mode): $f"; exit 1; fi
    done
    echo "$mode: clean"
done
