"""Device inflate alone: a C2-shape BAM's BGZF blocks in launches of N blocks, decoded GB/s of k_bgzf_inflate (HIP events around the
kernel, mm_bgzf_times) and of the CRC kernel.  python tools/inflate_bench.py [reads] [blocks per launch]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from minimod_amd import bgzf, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ref = synth.reference(3, 16 << 20)
bs = [synth.batch(ref, i * 2048, 2048, seed=9, n_reads_total=reads, with_order=False) for i in range(reads // 2048)]
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "s.bam")
    synth.write_bam_parallel(path, [("chrS", len(ref))], bs, threads=8)
    blocks = bgzf.split_bgzf(open(path, "rb").read())
inf = bgzf.Inflater(slots=1, max_blocks=per, max_cbytes=per * 66000, max_obytes=per * 65536)
res = []
for rep in range(3):
    for k in range(0, len(blocks) - per + 1, per):
        n, c, o = inf.fill(0, blocks[k:k + per])
        inf.submit(0, n, c, o)
        st = inf.wait(0, n)
        assert os.environ.get("MM_BENCH_NOCHECK") or not st.any(), st[st != 0][:5]
        t = inf.times(0)
        res.append((o / t["inflate_ms"] / 1e6, o / t["crc_ms"] / 1e6, t["inflate_ms"], o, c))
res = res[len(res) // 3:]
print("launches of %d blocks: inflate %.1f GB/s decoded (min %.1f, max %.1f), CRC %.1f GB/s; a launch: %.2f ms for %.1f MB decoded from %.1f MB" % (
    per, np.median([r[0] for r in res]), min(r[0] for r in res), max(r[0] for r in res), np.median([r[1] for r in res]), np.median([r[2] for r in res]),
    res[0][3] / 1e6, res[0][4] / 1e6))
