import sys, time
sys.path.insert(0, '.')
import numpy as np
import minimod_amd
from minimod_amd import synth, engine
ref = synth.reference(1, 50 << 20)
for first in (0, 4096*7):
    b = synth.batch(ref, first, 4096, seed=5, n_reads_total=100000)
    eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)])
    eng.stats_enable(True)
    t = eng.submit(b); eng.wait(t)
    st = eng.stats_get()
    ph = st["phase_cycles"]
    print("max L", int(b["reads"]["l_qseq"].max()), "max ncig", int(b["reads"]["n_cigar"].max()), "KA item max us: cigar, mm, dir", [round(x/100,1) for x in ph[0:3]], "items", ph[3])
    eng.close()
