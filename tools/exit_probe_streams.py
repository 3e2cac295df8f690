"""what HIP streams cost a process: tools/exit_probe_streams.hip over stream counts"""
import subprocess, sys, time
exe = sys.argv[1]
for args in (["0", "0"], ["1", "0"], ["2", "0"], ["4", "0"], ["8", "0"], ["16", "0"], ["3", "1"], ["6", "1"], ["12", "1"], ["8", "0", "destroy"], ["12", "1", "destroy"]):
    best = None
    for _ in range(3):
        r = subprocess.run([exe] + args, stdout=subprocess.PIPE)
        e1 = time.time()
        v = [float(x) for x in r.stdout.split()]
        if best is None or e1 - v[0] < best[0]:
            best = (e1 - v[0], v)
    d, v = best
    print("%2s streams %s %s: made in %.3f s, first launches %.3f s, resident %.2f -> %.2f GB, destroyed in %.3f s, last word to reaped %.3f s" % (args[0], "priorities" if args[1] == "1" else "plain     ", "destroyed" if len(args) > 2 else "held     ", v[1], v[2], v[3], v[4], v[5], d))
