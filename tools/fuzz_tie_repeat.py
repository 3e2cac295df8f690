"""one seed of tools/fuzz_tie.py again and again (a replay that gives up once in a while): python tools/fuzz_tie_repeat.py <seed> <times>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]] + sys.argv[1:]
seed, times = int(sys.argv[1]), int(sys.argv[2])
import importlib.util
spec = importlib.util.spec_from_file_location("fuzz_tie", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_tie.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
bad = 0
for i in range(times):
    out = m.run(seed)
    if out[0] != "same":
        bad += 1
        print("try", i, out, flush=True)
print("seed %d: %d of %d tries not the same" % (seed, bad, times))
