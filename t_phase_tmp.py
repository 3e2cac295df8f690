import sys, time
sys.path.insert(0, '.')
import numpy as np
import minimod_amd
from minimod_amd import synth
ref = synth.reference(1, 50 << 20)
b = synth.batch(ref, 0, 4096, seed=5, n_reads_total=100000, max_len=float(sys.argv[1]) if len(sys.argv) > 1 else 0.0)
eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)])
eng.stats_enable(True)
for rep in range(2):
    t = eng.submit(b, b["order"]); eng.wait(t)
    st = eng.stats_get()
    ph = st["phase_cycles"]
    print("kernel ms", round(eng.kernel_ms(t),3), "KA phases us/read (100MHz ticks): pass1, reserve, cigar, dir, pass2", [round(x/4096/100,2) for x in ph[4:9]])
