"""tools/churn_probe.hip under process churn: W workers start it again and again.  usage: python tools/churn_probe.py <exe> <workers> <runs each> <GB>"""
import subprocess, sys, time
from concurrent.futures import ThreadPoolExecutor
exe, W, K, gb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
def worker(w):
    bad = []
    for k in range(K):
        r = subprocess.run([exe, gb, "15"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            bad.append((r.returncode, (r.stdout + r.stderr).decode(errors="replace")[-300:]))
    return bad
t0 = time.time()
with ThreadPoolExecutor(max_workers=W) as ex:
    res = [b for bl in ex.map(worker, range(W)) for b in bl]
print("%d workers x %d processes holding %s GB each in %.0f s: %d bad %s" % (W, K, gb, time.time() - t0, len(res), [r[0] for r in res]))
for rc, out in res[:4]:
    print("----", rc, out)
