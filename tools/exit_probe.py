"""time from a process's last word to its being reaped, by what it holds (tools/exit_probe.hip)"""
import subprocess, sys, time
exe = sys.argv[1]
for args in (["0", "0", "0"], ["1", "0", "0"], ["4", "0", "0"], ["8", "0", "0"], ["8", "0", "0", "free"], ["0", "200", "0"], ["0", "800", "0"], ["0", "800", "0", "free"], ["0", "0", "2"], ["0", "0", "6"], ["8", "400", "3"]):
    best = None
    for _ in range(3):
        r = subprocess.run([exe] + args, stdout=subprocess.PIPE)
        e1 = time.time()
        last, freed = [float(x) for x in r.stdout.split()]
        best = min(best, e1 - last) if best is not None else e1 - last
    print("device GB %s pinned MB %s host GB %s %s: %.3f s from the last word to reaped (explicit frees %.3f s)" % (args[0], args[1], args[2], "freed" if len(args) > 3 else "held ", best, freed))
