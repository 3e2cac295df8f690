"""End-to-end timing of the CLI (minimod_amd/bin/minimod freq|view) on a synthetic BGZF BAM + FASTA written to /tmp.
Usage: python tools/e2e_cli.py [n_batches_of_4096_reads]   (needs a GPU; numbers quoted in DESIGN.md section 5)."""
import sys, os, time, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
import numpy as np
from minimod_amd import synth
ref = synth.reference(3, 48 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=4096 * NB) for i in range(NB)]
bases = sum(b["n_bases"] for b in bs)
os.makedirs('/tmp/e2e', exist_ok=True)
synth.write_bam('/tmp/e2e/s.bam', [("chrS", len(ref))], bs)
synth.write_fasta('/tmp/e2e/s.fa', "chrS", ref)
print("BAM MB", os.path.getsize('/tmp/e2e/s.bam') / 1e6, "Mbases", bases / 1e6, "cores", os.cpu_count())
BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'minimod_amd', 'bin', 'minimod')
for tool, extra in (("freq", ["-b", "-m", "0.8"]), ("view", [])):
    for t in (8, 32, 64):
        t0 = time.time()
        r = subprocess.run([BIN, tool, "-c", "m[CG]", "-K", "4096", "-B", "200M", "-t", str(t), "-o", "/tmp/e2e/out.txt"] + extra + ["/tmp/e2e/s.fa", "/tmp/e2e/s.bam"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.time() - t0
        err = r.stderr.decode()
        keys = [l.split("] ")[-1] for l in err.splitlines() if "time:" in l or "loaded in" in l]
        print(tool, "-t", t, "rc", r.returncode, "wall %.3f s -> %.0f Mbases/s" % (dt, bases / dt / 1e6), "|", "; ".join(keys), "| out MB", os.path.getsize('/tmp/e2e/out.txt') / 1e6)
