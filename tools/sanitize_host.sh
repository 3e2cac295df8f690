#!/bin/bash
# Build the C ingestion path (bamio.c + loader.c) and the row formatter (emit.c) with AddressSanitizer+UBSan and with
# ThreadSanitizer; run the former over the bundled BAMs (and any BAMs given as arguments), the latter over synthetic rows
# (pool-parallel pieces + writer thread vs the serial run); the DEFLATE decoder (inflate_fast.c) on round trips and 20 000 damaged
# streams under ASan+UBSan; the mapped FASTA parser against the stream parser.  CPU only: GPU sanitizers are not available on this pool.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
H="$here/../minimod_amd/csrc/host"
out="${TMPDIR:-/tmp}/mm_sanitize"
mkdir -p "$out"
for mode in address,undefined thread; do
    bin="$out/loader_bench_${mode%%,*}"
    gcc -O1 -g -fsanitize=$mode -fno-omit-frame-pointer -std=gnu99 -I"$H" -I"$here/../include" -o "$bin" "$here/loader_bench.c" "$H/loader.c" "$H/bamio.c" "$H/inflate_fast.c" "$H/crc32_fast.c" -lz -lpthread
    for f in "$here"/../tests/golden/data/*.bam "$@"; do
        ASAN_OPTIONS=detect_leaks=1 "$bin" "$f" 5 > "$out/last.log" 2>&1 || { cat "$out/last.log"; echo "FAILED ($mode): $f"; exit 1; }
        if grep -q "ERROR: \|WARNING: ThreadSanitizer\|runtime error" "$out/last.log"; then cat "$out/last.log"; echo "REPORT ($mode): $f"; exit 1; fi
    done
    ebin="$out/emit_check_${mode%%,*}"
    gcc -O1 -g -fsanitize=$mode -fno-omit-frame-pointer -std=gnu11 -I"$H" -I"$here/../include" -o "$ebin" "$here/emit_check.c" "$H/emit.c" "$H/loader.c" "$H/bamio.c" "$H/inflate_fast.c" "$H/crc32_fast.c" -lz -lpthread
    ASAN_OPTIONS=detect_leaks=1 "$ebin" 300000 6 "$out/emit.txt" > "$out/last.log" 2>&1 || { cat "$out/last.log"; echo "FAILED ($mode): emit_check"; exit 1; }
    if grep -q "ERROR: \|WARNING: ThreadSanitizer\|runtime error" "$out/last.log"; then cat "$out/last.log"; echo "REPORT ($mode): emit_check"; exit 1; fi
    if [ "$mode" != thread ]; then
        ibin="$out/inflate_check_${mode%%,*}"
        gcc -O1 -g -fsanitize=$mode -fno-omit-frame-pointer -std=gnu11 -I"$H" -o "$ibin" "$here/inflate_check.c" "$H/inflate_fast.c" -lz
        ASAN_OPTIONS=detect_leaks=1 "$ibin" 20000 > "$out/last.log" 2>&1 || { cat "$out/last.log"; echo "FAILED ($mode): inflate_check"; exit 1; }
        if grep -q "ERROR: \|runtime error" "$out/last.log"; then cat "$out/last.log"; echo "REPORT ($mode): inflate_check"; exit 1; fi
    fi
    sbin="$out/share_check_${mode%%,*}"
    gcc -O1 -g -fsanitize=$mode -fno-omit-frame-pointer -std=gnu99 -I"$H" -I"$here/../include" -o "$sbin" "$here/share_check.c" "$H/loader.c" "$H/bamio.c" "$H/inflate_fast.c" "$H/crc32_fast.c" -lz -lpthread
    for f in "$@"; do
        if [ -f "$f.bai" ]; then
            ASAN_OPTIONS=detect_leaks=1 "$sbin" "$f" 1048576 > "$out/last.log" 2>&1 || { cat "$out/last.log"; echo "FAILED ($mode): share_check $f"; exit 1; }
            if grep -q "ERROR: \|WARNING: ThreadSanitizer\|runtime error" "$out/last.log"; then cat "$out/last.log"; echo "REPORT ($mode): share_check"; exit 1; fi
        fi
    done
    fbin="$out/fasta_check_${mode%%,*}"
    gcc -O1 -g -fsanitize=$mode -fno-omit-frame-pointer -std=gnu99 -I"$H" -I"$here/../include" -o "$fbin" "$here/fasta_check.c" "$H/fasta.c" "$H/bamio.c" "$H/inflate_fast.c" "$H/crc32_fast.c" -lz -lpthread
    printf '>a d\nAC GT\r\n\n>b\nTT>x\n>c\n>d\nA' > "$out/t1.fa"; printf 'junk\n>a\nACGT\n>b' > "$out/t2.fa"
    python3 -c "import random; random.seed(1); open('$out/t3.fa','w').write('>big\n' + '\n'.join(''.join(random.choice('ACGTN') for _ in range(70)) for _ in range(150000)) + '\n>t\nAC\n')"
    ASAN_OPTIONS=detect_leaks=1 "$fbin" "$out/t1.fa" "$out/t2.fa" "$out/t3.fa" > "$out/last.log" 2>&1 || { cat "$out/last.log"; echo "FAILED ($mode): fasta_check"; exit 1; }
    if grep -q "ERROR: \|WARNING: ThreadSanitizer\|runtime error" "$out/last.log"; then cat "$out/last.log"; echo "REPORT ($mode): fasta_check"; exit 1; fi
    echo "$mode: clean"
done
