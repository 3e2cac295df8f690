"""how long the CLI's exit helper (main.c, last_one_out) lives behind the process: the 1.5-Gbase file of tools/cli_trace_any.sh, five runs"""
import os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from minimod_amd import synth
ref = synth.reference(3, 48 << 20)
bs = [synth.batch(ref, i * 4096, 4096, seed=9, n_reads_total=49152, with_order=False) for i in range(12)]
os.makedirs("/tmp/ehl", exist_ok=True)
synth.write_bam_parallel("/tmp/ehl/s.bam", [("chrS", len(ref))], bs, threads=8)
synth.write_fasta("/tmp/ehl/s.fa", "chrS", ref)
cli = os.path.join(root, "minimod_amd", "bin", "minimod")
def helpers():
    n = 0
    for p in os.listdir("/proc"):
        if p.isdigit():
            try:
                with open("/proc/%s/stat" % p) as f:
                    s = f.read()
                if "(minimod)" in s and s.split(") ")[1][0] != "Z":
                    n += 1
            except OSError:
                pass
    return n
for i in range(6):
    env = dict(os.environ, MM_SYNC_EXIT="1") if i % 3 == 2 else os.environ
    t0 = time.time()
    r = subprocess.run([cli, "freq", "-b", "-c", "m[CG]", "-t", "16", "--gpu-ingest", "-o", "/tmp/ehl/o.bed", "/tmp/ehl/s.fa", "/tmp/ehl/s.bam"], stderr=subprocess.PIPE, env=env)
    t1 = time.time()
    while helpers() and time.time() - t1 < 20:
        time.sleep(0.002)
    t2 = time.time()
    print("run %d (%s): rc %d, wall %.3f s, the helper gone %.3f s later" % (i, "sync exit" if i % 3 == 2 else "helper", r.returncode, t1 - t0, t2 - t1), flush=True)
    time.sleep(0.2 if i == 3 else 2.0)
