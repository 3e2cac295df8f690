#!/bin/bash
# whole-process wall of `minimod freq` by input size, device ingestion against the host reader: tools/ingest_threshold.sh <batches of 4096 reads>...
root=$(cd "$(dirname "$0")/.." && pwd)
for nb in "$@"; do
python3 - $nb <<PY
import os, sys
sys.path.insert(0, "$root")
from minimod_amd import synth
nb = int(sys.argv[1])
ref = synth.reference(3, max(4 << 20, nb * 4 << 20))
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=nb * 4096, with_order=False) for i in range(nb)]
os.makedirs("/tmp/r4thr", exist_ok=True)
synth.write_bam_parallel("/tmp/r4thr/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/r4thr/s.fa", "chrS", ref)
PY
  python3 - $root <<'PY'
import os, re, subprocess, sys, time
root = sys.argv[1]
sz = os.path.getsize("/tmp/r4thr/s.bam")
for fl in ("--gpu-ingest", "--no-gpu-ingest"):
    ts, inner, last = [], [], ""
    for i in range(4):
        t = time.perf_counter()
        r = subprocess.run([root + "/minimod_amd/bin/minimod", "freq", "-b", "-c", "m[CG]", "-t", "16", fl, "-o", "/tmp/r4thr/o%s.bed" % fl, "/tmp/r4thr/s.fa", "/tmp/r4thr/s.bam"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True, env=dict(os.environ, MM_TIMELINE="1"))
        ts.append(time.perf_counter() - t)
        m = re.search(r"Real time: ([0-9.]+) sec", r.stderr.decode())
        inner.append(float(m.group(1)) if m else -1.0)
        if i == 3: last = r.stderr.decode()
    print("BAM %5d MiB  %-16s best of 4: %.3f s  (%s); the process's own clock: %s" % (sz >> 20, fl, min(ts), " ".join("%.3f" % x for x in ts), " ".join("%.3f" % x for x in inner)))
    if os.environ.get("MM_SHOW_TIMELINE"): print("\n".join(l for l in last.splitlines() if "timeline" in l or "runtime ready" in l))
same = open("/tmp/r4thr/o--gpu-ingest.bed", "rb").read() == open("/tmp/r4thr/o--no-gpu-ingest.bed", "rb").read()
print("   same bytes" if same else "   DIFFERENT OUTPUT")
PY
  rm -rf /tmp/r4thr
done
