import sys, time
sys.path.insert(0, '.')
import numpy as np
import minimod_amd
from minimod_amd import synth, engine
ref = synth.reference(1, 50 << 20)
b = synth.batch(ref, 0, 4096, seed=5, n_reads_total=100000)
eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)])
eng.stats_enable(True)
for rep in range(2):
    t = eng.submit(b); eng.wait(t)
    st = eng.stats_get()
    ph = st["phase_cycles"][4:9]; tot = sum(ph) or 1
    print("path ms", round(eng.kernel_ms(t),3), "KC phases % (ctx+carries, mm parse, ranks+dir slice, cig slice, calls):", [round(100*x/tot,1) for x in ph], "us per wave total", round(tot/4096/100,1))
