"""mm_freq_submit with opts.coalesce > 1 (the product's entry point, process_db's place, reference src/minimod.c:344-350):
consecutive HOST batches are staged one behind the other in device memory and launched together.  Same rows as the oracle
whatever the grouping; errors named relative to the group's first read; the CLI reaches k_stream_reads this way."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "minimod_amd", "bin", "minimod")

HOST_WORKER = r'''
import json, sys
sys.path.insert(0, %r)
STREAM_MODE = %d
import numpy as np
import minimod_amd
from minimod_amd import synth
from oracle import oracle as O
ref = synth.reference(21, 4 << 20)
sizes_in = [300, 300, 17, 300, 1, 300, 257]
bs, first = [], 0
for n in sizes_in:
    bs.append(synth.batch(ref, first, n, seed=9, n_reads_total=sum(sizes_in), with_order=False)); first += n
mods, th = [("m", "CG"), ("h", "CG")], [0.8, 0.7]
key = lambda r: list(zip(r["pos"].tolist(), r["strand"].tolist(), r["code"].tolist(), r["n_called"].tolist(), r["n_mod"].tolist()))
out = {}
for kw_name, kw in (("plain", {}), ("ins_hap", dict(insertions=True, haplotypes=True))):
    orc = O.Oracle(mods, th, ["chrS"], **kw); orc.add_contig("chrS", ref)
    for b in bs: orc.process(b, threads=8)
    want = orc.rows()
    for name, coalesce, mb in (("off", 1, 0), ("groups_of_3", 3, 0), ("one_group", 16, 0), ("small_staging", 16, 8)):
        eng = minimod_amd.FreqEngine([(c, x, t) for (c, x), t in zip(mods, th)], [("chrS", len(ref), ref)], coalesce=coalesce, gather_mb=mb, stream_mode=STREAM_MODE, **kw)
        tickets = []
        for b in bs:
            t = eng.submit(b); tickets.append(t); eng.host_done(t)
            for k in ("reads", "cigar", "seq", "mm", "ml"): pass
        sizes = {}
        for t in tickets: sizes[t] = eng.ticket_batches(t)
        got = eng.finalize(); lc = eng.launch_counts(); eng.close()
        a = sorted(zip(got["pos"].tolist(), got["strand"].tolist(), got["code"].tolist(), got["ins_offset"].tolist(), got["hp"].tolist(), got["n_called"].tolist(), got["n_mod"].tolist()))
        w = sorted(zip(want["pos"].tolist(), want["strand"].tolist(), want["code"].tolist(), want["ins_off"].tolist(), want["hp"].tolist(), want["n_called"].tolist(), want["n_mod"].tolist()))
        out[kw_name + ":" + name] = {"equal": a == w, "rows": len(w), "tickets": tickets, "launches": lc}
# the host memory may be reused as soon as host_done returns: scribble over every batch right after its submit
eng = minimod_amd.FreqEngine([("m", "CG", 0.8), ("h", "CG", 0.7)], [("chrS", len(ref), ref)], coalesce=16, stream_mode=STREAM_MODE)
for b in bs:
    c = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in b.items()}
    t = eng.submit(c); eng.host_done(t)
    for k in ("reads", "cigar", "seq", "mm", "ml"): c[k].view(np.uint8)[:] = 0xA5
got = eng.finalize(); eng.close()
orc = O.Oracle(mods, th, ["chrS"]); orc.add_contig("chrS", ref)
for b in bs: orc.process(b, threads=8)
out["scribbled"] = key(got) == key(orc.rows())
# a failing read inside a gathered group: index counted from the group's first read, its record on request
bad = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in bs[3].items()}
r = bad["reads"][150]; bad["cigar"][int(r["cigar_off"])] = (5 << 4) | 5   # a hard clip
eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)], coalesce=8, stream_mode=STREAM_MODE)
ts = [eng.submit(b) for b in bs[:3]] + [eng.submit(bad)]
try:
    eng.wait(ts[0]); out["error"] = None
except minimod_amd.engine.MinimodHipError as e:
    rec = eng.read_record(ts[0], e.read)
    out["error"] = [e.code, e.read, int(rec["pos"]) == int(r["pos"]) and int(rec["l_qseq"]) == int(r["l_qseq"]), len(set(ts))]
eng.close()
print(json.dumps(out))
'''


@pytest.mark.parametrize("stream_mode", [2, 1], ids=["stream", "tiles"])
def test_host_batches_gathered_into_launches(stream_mode):
    r = subprocess.run([sys.executable, "-c", HOST_WORKER % (ROOT, stream_mode)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    for kw in ("plain", "ins_hap"):
        for name in ("off", "groups_of_3", "one_group", "small_staging"):
            assert res[kw + ":" + name]["equal"] and res[kw + ":" + name]["rows"] > 1000, (kw, name)
        assert res[kw + ":off"]["launches"]["launches"] == 7
        t3 = res[kw + ":groups_of_3"]["tickets"]
        assert t3[0] == t3[1] == t3[2] != t3[3] and t3[3] == t3[5] != t3[6] and res[kw + ":groups_of_3"]["launches"]["launches"] == 3
        assert len(set(res[kw + ":one_group"]["tickets"])) == 1 and res[kw + ":one_group"]["launches"]["launches"] == 1
        assert 1 < res[kw + ":small_staging"]["launches"]["launches"] < 7     # the staging budget, not the count, ended the groups
    if stream_mode == 2:
        assert res["plain:one_group"]["launches"]["stream_launches"] == 1 and res["ins_hap:one_group"]["launches"]["stream_launches"] == 1   # (--insertions / --haplotypes stream too: the kIns instantiation)
    assert res["scribbled"]
    assert res["error"] == [1, 300 + 300 + 17 + 150, True, 1]


WILD_WORKER = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np
import torch
torch.zeros(1, device="cuda")
import minimod_amd
from oracle import oracle as O
from oracle import pybam
from tests.cases import KAT_REF, KAT_SEQ
# -c '*': window 2 of a gathered group carries a code no earlier read had (interned after the group began)
w1 = [pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+m?,0,0;", [255, 3]), pybam.make_record(0, 2, 16, KAT_SEQ, "20M", "C+m?,1;", [250])]
w2 = [pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+h?,0,1;", [255, 2]), pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+76792?,0;A+a?,0;", [9, 255])]
w3 = [pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+m?,0;C+h?,2;", [1, 255])]
out = {}
orc = O.Oracle([("*", "*")], [0.8], ["chrT"]); orc.add_contig("chrT", KAT_REF.encode())
for w in (w1, w2, w3): orc.process(pybam.flatten(w))
want = orc.rows(); wn = orc.code_names()
W = sorted((int(r["pos"]), int(r["strand"]), wn[r["code"]], int(r["n_called"]), int(r["n_mod"])) for r in want)
# host batches
eng = minimod_amd.FreqEngine([("*", "*", 0.8)], [("chrT", len(KAT_REF), KAT_REF.encode())], coalesce=8, stream_mode=2)
ts = [eng.submit(pybam.flatten(w)) for w in (w1, w2, w3)]
got = eng.finalize(); gn = eng.code_names(); lc = eng.launch_counts(); eng.close()
out["host"] = sorted((int(r["pos"]), int(r["strand"]), gn[r["code"]], int(r["n_called"]), int(r["n_mod"])) for r in got) == W
out["host_launches"] = lc["launches"]
# windows of one resident read set
b = pybam.flatten(w1 + w2 + w3)
dev = {k: torch.from_numpy(b[k].view(np.uint8).reshape(-1)).cuda() for k in ("reads", "cigar", "seq", "mm", "ml")}
def window(lo, hi):
    return dict(reads=dev["reads"].data_ptr() + 64 * lo, cigar=dev["cigar"].data_ptr(), seq=dev["seq"].data_ptr(), mm=dev["mm"].data_ptr(), ml=dev["ml"].data_ptr(),
                n_reads=hi - lo, n_cigar_words=len(b["cigar"]), n_seq_bytes=len(b["seq"]), n_mm_bytes=len(b["mm"]), n_ml_bytes=len(b["ml"]),
                max_n_cigar=int(b["reads"]["n_cigar"].max()), max_l_qseq=int(b["reads"]["l_qseq"].max()))
eng = minimod_amd.FreqEngine([("*", "*", 0.8)], [("chrT", len(KAT_REF), KAT_REF.encode())], coalesce=8, stream_mode=2)
lo = 0
for w in (w1, w2, w3):
    eng.intern_codes_from(pybam.flatten(w))       # the CLI's protocol: intern, then submit
    eng.submit_device(window(lo, lo + len(w))); lo += len(w)
got = eng.finalize(); gn = eng.code_names(); eng.close()
out["device"] = sorted((int(r["pos"]), int(r["strand"]), gn[r["code"]], int(r["n_called"]), int(r["n_mod"])) for r in got) == W
out["rows"] = len(W)
print(json.dumps(out))
'''


def test_wildcard_code_interned_inside_a_gathered_group():
    """-c '*' with coalesce > 1: a later window brings a code string the device's table does not hold yet; the group ends there
    (the table is uploaded in front of a launch's first window) and the rows equal the oracle's."""
    r = subprocess.run([sys.executable, "-c", WILD_WORKER % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert res["host"] and res["device"] and res["rows"] >= 6 and res["host_launches"] >= 2, res


def test_cli_batches_reach_the_streaming_kernel(tmp_path):
    """`minimod freq` with the reference's default -K 512 on a BAM big enough for the default routing: the CLI's batches are
    gathered into a few launches and k_stream_reads runs in them (nothing in the command asks for it); bytes equal the
    oracle's; `--gather 1` (every batch its own launch) prints the same."""
    from minimod_amd import synth
    from oracle import oracle as O
    import minimod_amd
    minimod_amd.build_all()
    ref = synth.reference(33, 8 << 20)
    n = 9600
    bs = [synth.batch(ref, i, min(1200, n - i), seed=5, n_reads_total=n, with_order=False) for i in range(0, n, 1200)]
    bam, fa = str(tmp_path / "r.bam"), str(tmp_path / "r.fa")
    synth.write_bam(bam, [("chrS", len(ref))], bs)
    synth.write_fasta(fa, "chrS", ref)
    cmd = [BIN, "freq", "-b", "-c", "m[CG]", "-m", "0.8", "-t", "8", "-B", "100M"]
    r = subprocess.run(cmd + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    err = r.stderr.decode()
    import re
    m = re.search(r"GPU launches: (\d+) for (\d+) batches \((\d+) with k_stream_reads\)", err)
    assert m, err[-1500:]
    launches, batches, streamed = (int(x) for x in m.groups())
    # -K 512 does not make the launches small: the batches are gathered by what the staging takes (1 GiB), and every launch streams
    assert batches >= 19 and launches == 1 and streamed == launches, (launches, batches, streamed)
    orc = O.Oracle([("m", "CG")], [0.8], ["chrS"])
    orc.add_contig("chrS", ref)
    for b in bs:
        orc.process(b, threads=8)
    want = O.format_rows(orc.rows(), ["chrS"], orc.code_names(), bedmethyl=True)
    assert len(want) > 100000 and r.stdout.decode() == want
    r1 = subprocess.run(cmd + ["--gather", "1", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r1.returncode == 0 and r1.stdout == r.stdout
    m1 = re.search(r"GPU launches: (\d+) for (\d+) batches \((\d+) with", r1.stderr.decode())
    assert m1 and int(m1.group(1)) == int(m1.group(2))
    # `minimod view` gathers as well (round 4): its launches stream, its rows are those of one launch per batch
    vcmd = [BIN, "view", "-c", "m[CG]", "-t", "8", "-B", "100M"]
    v = subprocess.run(vcmd + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert v.returncode == 0, v.stderr.decode()[-2000:]
    mv = re.search(r"GPU launches: (\d+) for (\d+) batches \((\d+) with k_stream_reads\)", v.stderr.decode())
    assert mv and int(mv.group(1)) <= 2 and int(mv.group(3)) == int(mv.group(1)), v.stderr.decode()[-800:]
    v1 = subprocess.run(vcmd + ["--gather", "1", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert v1.returncode == 0 and v1.stdout == v.stdout and len(v.stdout) > 1000000


def test_cli_names_a_failing_read_of_a_gathered_group(tmp_path):
    """A hard clip in a read of the third -K batch: the CLI prints the reference's message with the read's index in ITS batch
    and its contig and position, and exits 1 (src/mod.c:841-844)."""
    from minimod_amd import synth
    from oracle import pybam
    from tests.cases import KAT_REF, KAT_SEQ
    recs = [pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+m?,0,0;", [255, 3]) for _ in range(11)]
    recs[9] = pybam.make_record(0, 2, 0, KAT_SEQ, "3H20M", "C+m?,0,0;", [255, 3])
    bam, fa = str(tmp_path / "h.bam"), str(tmp_path / "h.fa")
    synth.write_bam(bam, [("chrT", len(KAT_REF))], [pybam.flatten(recs)], filter_fodder=False)
    synth.write_fasta(fa, "chrT", np.frombuffer(KAT_REF.encode(), dtype=np.uint8))
    r = subprocess.run([BIN, "freq", "-K", "4", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 1
    assert b"Hard clipping found in read 1 of the batch (contig chrT, pos 2)" in r.stderr, r.stderr.decode()[-1500:]
    # the device-side reader's batches are not -K batches: the message still names the read by its place in the reference's batch
    # (round 5: the host reader walks the file up to it on the error path), for two batchings, tied runs (device replay) included
    for extra, want in ((["-K", "4"], b"read 1 of the batch"), (["-K", "7"], b"read 2 of the batch"), (["-K", "4", "-c", "m,h"], b"read 1 of the batch")):
        r = subprocess.run([BIN, "freq", "--gpu-ingest"] + extra + [fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 1 and b"Hard clipping found in " + want + b" (contig chrT, pos 2)" in r.stderr, r.stderr.decode()[-1500:]
