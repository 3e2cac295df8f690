import sys, time
sys.path.insert(0, '.')
import numpy as np
import minimod_amd
from minimod_amd import synth
ref = synth.reference(1, 50 << 20)
b = synth.batch(ref, 0, 4096, seed=5, n_reads_total=100000)
eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)])
eng.stats_enable(True)
for rep in range(2):
    t = eng.submit(b, b["order"]); eng.wait(t)
    st = eng.stats_get()
    ph = st["phase_cycles"]; tot = sum(ph) or 1
    print("kernel ms", eng.kernel_ms(t), "phases cigar/dir/mmparse/flush %", [round(100*x/tot,1) for x in ph], "cycles/read", [x//4096 for x in ph], st["lookups"], st["dense_updates"])
