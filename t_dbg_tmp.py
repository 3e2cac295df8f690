import sys
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O, pybam
import minimod_amd
from minimod_amd import synth
from tests.hiprun import hip_rows_from_records
from tests.cases import KAT_REF, kat_records, KAT_M
print('KAT', [r[:4] for r in hip_rows_from_records(kat_records(), KAT_REF, "m")] == KAT_M)
print([r[:4] for r in hip_rows_from_records(kat_records(), KAT_REF, "m")])
ref = synth.reference(1, 4 << 20)
for n in (1, 8, 200):
    b = synth.batch(ref, 0, n, seed=5, n_reads_total=2000, max_len=30000.0)
    eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)])
    eng.stats_enable(True)
    eng.process(b)
    got = eng.finalize(); st = eng.stats_get(); eng.close()
    o = O.Oracle([("m", "CG")], [0.8], ["chrS"]); o.add_contig("chrS", ref); o.process(b); want = o.rows()
    gs = set(zip(got["pos"].tolist(), got["strand"].tolist(), got["n_called"].tolist(), got["n_mod"].tolist()))
    ws = set(zip(want["pos"].tolist(), want["strand"].tolist(), want["n_called"].tolist(), want["n_mod"].tolist()))
    print(n, 'rows', len(got), len(want), 'missing', len(ws - gs), 'extra', len(gs - ws), st)
    if ws != gs:
        print(' first missing', sorted(ws - gs)[:5], 'first extra', sorted(gs - ws)[:5], 'reads', b['reads'][['pos','l_qseq','flag','mm_len']][:3])
