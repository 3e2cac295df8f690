import sys, time
sys.path.insert(0, '.')
from oracle import oracle as O, pybam
import minimod_amd
from tests.hiprun import hip_rows_from_records
from tests.cases import KAT_REF, kat_records
which = sys.argv[1]
recs = {
 'x1': [pybam.make_record(0, 2, 0, "CGTT", "4M", "X+m?,0;", [255])],
 'x5': kat_records()[:2] + [pybam.make_record(0, 2, 0, "CGTT", "4M", "X+m?,0;", [255])] + kat_records()[2:],
 's1': [pybam.make_record(0, 2, 0, "CGTT", "4M", "C*m?,0;", [255])],
 'r1': [pybam.make_record(0, 2, 0, "CGTT", "4M", "C+m?,0,0;", [255])],
 'l1': [pybam.make_record(0, 2, 0, "CGTT", "4M", "C+m?,1234567890;", [255])],
 'm1': [pybam.make_record(0, 2, 0, "CGTTCG", "6M", "C+m?,0,0;", [255])],
 'p1': [pybam.make_record(0, 28, 0, "CGTT", "4M", "C+m?,0;", [255])],
}[which]
try:
    print(which, hip_rows_from_records(recs, KAT_REF, "m"))
except minimod_amd.MinimodHipError as e:
    print(which, 'error', e.code, e.read)
