"""Both device forms of the hot path -- the tile pipeline (default) and the fused one-wave-per-read kernel
(opts.force_fused, also the fallback for reads the tiles do not cover) -- against the oracle on synthetic reads that
exercise what the bundled BAMs do not at scale: long reads (spills past the LDS caps, many tiles), HiFi shape, '.'
groups (implicit calls, tail tiles), haplotypes, insertions, the side list."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKER = r'''
import json, sys
sys.path.insert(0, %r)
FUSED = bool(%d)
STREAM_MODE = %d
import numpy as np
import minimod_amd
from minimod_amd import synth
from oracle import oracle as O
ref = synth.reference(21, 6 << 20)
out = {}
cases = {
  "ont_long": dict(gen=dict(n=400, max_len=0.0), c=[("m","CG"),("h","CG")], th=[0.8,0.7], kw={}),
  "hifi_dot": dict(gen=dict(n=200, shape=1, dot_fraction=0.6), c=[("m","CG")], th=[0.8], kw={}),
  "dot_hp_ins": dict(gen=dict(n=150, dot_fraction=1.0, haplotypes=True, long_insertions=True, max_len=20000.0), c=[("m","C")], th=[0.7],
                     kw=dict(insertions=True, haplotypes=True)),
  "star_ctx_single": dict(gen=dict(n=200, single_code=True), c=[("m","*")], th=[0.9], kw={}),
  # round 4: '.' groups under --insertions / --haplotypes in k_stream_reads -- HiFi-shape reads (their whole CIGAR is in the
  # kernel's table: reverse reads take the mirrored anchor of mod.c:1234, :1314 there), and --haplotypes alone on ONT shapes
  "hifi_dot_hp_ins": dict(gen=dict(n=300, shape=1, dot_fraction=1.0, haplotypes=True, long_insertions=True), c=[("m","C")], th=[0.7],
                          kw=dict(insertions=True, haplotypes=True)),
  "dot_hp_only": dict(gen=dict(n=150, dot_fraction=1.0, haplotypes=True, max_len=20000.0), c=[("m","CG")], th=[0.8], kw=dict(haplotypes=True)),
  "dot_long": dict(gen=dict(n=60, dot_fraction=1.0), c=[("m","C")], th=[0.8], kw={}),   # view: tens of thousands of rows per read
  # BASELINE.json configs[2]: multi-mod -c m[CG],h[CG] -m 0.8,0.7 on PacBio-HiFi-shape reads with the MM '?' flag
  "hifi_q_multimod": dict(gen=dict(n=600, shape=1, dot_fraction=0.0), c=[("m","CG"),("h","CG")], th=[0.8,0.7], kw={}),
}
for name, cs in cases.items():
    g = dict(cs["gen"]); n = g.pop("n")
    b = synth.batch(ref, 0, n, seed=77, n_reads_total=n, **g)
    eng = minimod_amd.FreqEngine([(c, x, t) for (c, x), t in zip(cs["c"], cs["th"])], [("chrS", len(ref), ref)], force_fused=FUSED, stream_mode=STREAM_MODE, **cs["kw"])
    eng.stats_enable(True)
    eng.process(b)
    st = eng.stats_get()
    got = eng.finalize(); eng.close()
    orc = O.Oracle(cs["c"], cs["th"], ["chrS"], **cs["kw"]); orc.add_contig("chrS", ref); orc.process(b, threads=8)
    want = orc.rows()
    key = lambda r, io: sorted(zip(r["pos"].tolist(), r["strand"].tolist(), r["code"].tolist(), r[io].tolist(), r["hp"].tolist(), r["n_called"].tolist(), r["n_mod"].tolist()))
    out[name] = {"rows": int(len(want)), "equal": key(got, "ins_offset") == key(want, "ins_off"), "max_l": int(b["reads"]["l_qseq"].max()),
                 "to_tiles": st["stream_to_tiles"], "streamed": st["stream_done"], "rev_long": int(((b["reads"]["flag"] & 16) != 0).sum())}
    # the same batch through view mode: rows in print_view_output order, element for element
    eng = minimod_amd.FreqEngine([(c, x, t) for (c, x), t in zip(cs["c"], cs["th"])], [("chrS", len(ref), ref)], view=True, force_fused=FUSED, stream_mode=STREAM_MODE, **cs["kw"])
    v = eng.view(b); eng.close()
    orc = O.Oracle(cs["c"], cs["th"], ["chrS"], **cs["kw"]); orc.set_view(True); orc.add_contig("chrS", ref); orc.process(b, threads=8)
    w = orc.view_rows()
    vk = lambda r, io: list(zip(r["read"].tolist(), r["pos"].tolist(), r["read_pos"].tolist(), r["code"].tolist(), r[io].tolist(), r["prob"].tolist()))
    out[name]["view_rows"] = int(len(w)); out[name]["view_equal"] = vk(v, "ins_offset") == vk(w, "ins_off")
print(json.dumps(out))
'''


@pytest.mark.parametrize("fused,stream_mode", [(0, 2), (0, 3), (0, 1), (1, 1)], ids=["stream+tiles", "stream-dot+tiles", "tiles", "fused"])
def test_synthetic_shapes_match_oracle(fused, stream_mode):
    r = subprocess.run([sys.executable, "-c", WORKER % (ROOT, fused, stream_mode)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert set(res) == {"ont_long", "hifi_dot", "dot_hp_ins", "star_ctx_single", "dot_long", "hifi_q_multimod", "hifi_dot_hp_ins", "dot_hp_only"}
    if fused == 0 and stream_mode == 3:   # the '.'-capable instantiations from the first launch on
        assert res["hifi_dot_hp_ins"]["to_tiles"] == 0 and res["hifi_dot_hp_ins"]["streamed"] == 300, res["hifi_dot_hp_ins"]
        assert res["dot_hp_only"]["to_tiles"] == 0, res["dot_hp_only"]
        # ONT-shape reads under --insertions: the forward ones stream, the reverse ones of more than 512 ops are the tile pipeline's
        assert 0 < res["dot_hp_ins"]["streamed"] and res["dot_hp_ins"]["to_tiles"] <= res["dot_hp_ins"]["rev_long"], res["dot_hp_ins"]
    for name, v in res.items():
        assert v["rows"] > 1000, (name, v)
        assert v["equal"], (name, v)
        assert v["view_rows"] > 1000 and v["view_equal"], (name, v)
    assert res["ont_long"]["max_l"] > 40000


def _bench_batches(reads, batch, n):
    """The batches bench.py generates for rank 0 (same functions, same seed)."""
    import bench
    plan = bench.shard_plan(0, 1)
    ref = bench.gen_reference(plan, 0x5EED)
    return plan, ref, [bench.gen_batch(ref, plan, 0, 0x5EED, reads, batch, bi) for bi in range(n)]


def test_bench_freq_results_match_oracle(tmp_path):
    """bench.py on device-resident batches (mm_freq_submit_device); its --dump of the first two batches' rows is
    checked here against the oracle on the regenerated batches."""
    from oracle import oracle as O
    dump = str(tmp_path / "freq.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "6000", "--batch", "2048", "--steps", "6", "--warmup", "1",
                        "--dump", dump, "--cpu-sample-batches", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["unit"] == "Mbases/s" and d["value"] > 0 and 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["kind"] == "port"
    # the end-to-end leg: the CLI on the same reads as a BGZF BAM, its bedmethyl byte-identical to the CPU path's
    e2e, ce = d["end_to_end"], d["cpu_baseline_e2e"]
    assert e2e["reads"] == 6000 and e2e["value"] > 0 and e2e["parity_vs_cpu"]["byte_identical"] and e2e["parity_vs_cpu"]["bytes"] > 100000
    assert ce["kind"] == "port" and ce["t_all"]["value"] > 0 and ce["t_1"]["value"] > 0 and "process" in ce["t_1"]["stages_s"]
    assert "load" in e2e["stages_s"] and "device" in d["config"]["read_order"]
    got = np.load(dump)["rows"]
    plan, ref, batches = _bench_batches(6000, 2048, 2)
    orc = O.Oracle([("m", "CG")], [0.8], ["chrS"])
    orc.add_contig("chrS", ref)
    for hb in batches:
        orc.process(hb, threads=os.cpu_count() or 1)
    want = orc.rows()
    assert len(want) > 1000 and len(got) == len(want)
    for a, b in (("pos", "pos"), ("strand", "strand"), ("n_called", "n_called"), ("n_mod", "n_mod")):
        assert (got[a] == want[b]).all()


@pytest.mark.parametrize("config", ["C3", "C5"])
def test_bench_other_workloads_run(config):
    """bench.py --config C3 / C5 (BASELINE.json configs[2] and [4]) on a small read set: a line with a roofline object comes out."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--reads", "3000", "--batch", "1024", "--steps", "4",
                        "--warmup", "1", "--no-e2e", "--cpu-sample-batches", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["config"]["workload"].startswith(config) and d["value"] > 0 and 0 < d["roofline"]["frac"] < 1
    if config == "C5":
        assert d["roofline"]["side_list_updates_per_pass"] > 0


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset): bench.py starts the two ranks itself as fresh
    child processes (the parent never touches the GPU) and rank 0's line says n_gpus 2.  (gloo: the two ranks share the one GPU
    of the test box; with RCCL every rank needs a GPU of its own.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--reads", "6000", "--batch", "1024", "--steps", "4",
                        "--warmup", "1", "--reps", "2", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["final_reduce"]["ranks"] == 2 and d["final_reduce"]["backend"] == "gloo" and d["value"] > 0
    assert d["config"]["sharding"].startswith("24 contigs")
    fr = d["final_reduce"]
    assert fr["ranks_seen"] == 2 and len(fr["per_rank"]) == 2 and all(p["bases"] > 0 and p["Mbases_per_s"] > 0 for p in fr["per_rank"])
    assert sum(p["bases"] for p in fr["per_rank"]) > 0 and fr["slab_bytes"] > 0


def test_bench_config_c4_line():
    """`--config C4` (BASELINE.json configs[3]) scaled down: the 24-contig genome with hg38's length ratios at N = 1 as well, and
    with two ranks (gloo, sharing the GPU) the halo slab between them; the line names the workload."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--config", "C4", "--reads", "6000", "--region-mb", "24", "--batch", "1024", "--steps", "4", "--warmup", "1", "--reps", "2", "--no-cpu-baseline", "--no-extra"]
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "gloo"] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        d = json.loads([l for l in r.stdout.decode().strip().splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == n and d["config"]["workload"].startswith("C4:") and "24-contig genome" in d["config"]["workload"]
        assert d["config"]["sharding"].startswith("24 contigs") and d["value"] > 0
        if n == 2:
            assert d["final_reduce"]["ranks"] == 2 and len(d["final_reduce"]["per_rank"]) == 2


def test_bench_view_results_match_oracle(tmp_path):
    from oracle import oracle as O
    dump = str(tmp_path / "view.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "view", "--reads", "4096", "--batch", "2048", "--steps", "4",
                        "--warmup", "1", "--dump", dump], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["metric"] == "minimod view Mbases/sec" and d["value"] > 0
    got = np.load(dump)["rows"]
    plan, ref, batches = _bench_batches(4096, 2048, 1)
    orc = O.Oracle([("m", "CG")], [0.8], ["chrS"])
    orc.set_view(True)
    orc.add_contig("chrS", ref)
    orc.process(batches[0], threads=os.cpu_count() or 1)
    want = orc.view_rows()
    assert len(want) > 10000 and len(got) == len(want)
    for a, b in (("read", "read"), ("pos", "pos"), ("read_pos", "read_pos"), ("prob", "prob"), ("ins_offset", "ins_off")):
        assert (got[a] == want[b]).all()


DEPTH_WORKER = r'''
import json, sys, time
sys.path.insert(0, %r)
import numpy as np
import minimod_amd
from minimod_amd import synth
from oracle import oracle as O
# BASELINE.json configs[4]: --haplotypes --insertions, 200x depth on a 5 Mb region (per-HP planes, atomic contention)
region = 5 << 20
ref = synth.reference(31, region + (1 << 20))
n = 66000
bs = [synth.batch(ref, i * 4096, min(4096, n - i * 4096), seed=41, n_reads_total=n, region_begin=0, region_len=region, haplotypes=True)
      for i in range((n + 4095) // 4096)]
bases = sum(b["n_bases"] for b in bs)
eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)], insertions=True, haplotypes=True, side_capacity=32 << 20)
t0 = time.time()
tk = []
for b in bs:
    tk.append(eng.submit(b))
    if len(tk) >= 3: eng.wait(tk.pop(0))
for t in tk: eng.wait(t)
got = eng.finalize(); eng.close()
t_gpu = time.time() - t0
orc = O.Oracle([("m", "CG")], [0.8], ["chrS"], insertions=True, haplotypes=True); orc.add_contig("chrS", ref)
for b in bs: orc.process(b, threads=64)
want = orc.rows()
key = lambda r, io: (r["pos"].astype(np.int64) << 24) | (r["strand"].astype(np.int64) << 23) | ((r["hp"].astype(np.int64) + 1) << 17) | r[io].astype(np.int64)
ga, wa = np.argsort(key(got, "ins_offset"), kind="stable"), np.argsort(key(want, "ins_off"), kind="stable")
eq = len(got) == len(want) and (key(got, "ins_offset")[ga] == key(want, "ins_off")[wa]).all() and \
     (got["n_called"][ga] == want["n_called"][wa]).all() and (got["n_mod"][ga] == want["n_mod"][wa]).all()
print(json.dumps({"rows": int(len(want)), "equal": bool(eq), "depth": bases / region, "max_called": int(want["n_called"].max()), "gpu_s": t_gpu}))
'''


def test_deep_region_haplotypes_insertions():
    """200x depth on a 5 Mb region with --haplotypes --insertions (BASELINE.json configs[4]): every counter takes
    hundreds of atomic updates, the side list takes the inserted calls; bit-exact against the oracle."""
    r = subprocess.run([sys.executable, "-c", DEPTH_WORKER % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert res["depth"] > 150 and res["max_called"] > 60 and res["rows"] > 100000 and res["equal"], res


SHARD_WORKER = r'''
import json, sys
sys.path.insert(0, %r)
STREAM_MODE = %d
import torch
torch.zeros(1, device="cuda")   # torch's HIP runtime comes up before the library's (as in bench.py)
import minimod_amd
from minimod_amd import synth
from oracle import oracle as O
import bench
interval, halo = 1 << 20, 1 << 16
plans = [bench.shard_plan(r, 2, interval, halo) for r in range(2)]
ref = synth.reference(5, plans[0]["contig_len"])
batches = [synth.batch(ref, 0, 900, seed=11 + 7919 * r, contig_len=plans[r]["contig_len"], n_reads_total=900,
                       region_begin=plans[r]["read_begin"], region_len=plans[r]["read_len"], median_len=3000.0, max_len=30000.0)
           for r in range(2)]
engs = []
for r in range(2):
    e = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", plans[r]["contig_len"], ref)],
                               intervals=[(0, plans[r]["begin"], plans[r]["end"], plans[r]["halo"])], stream_mode=STREAM_MODE)
    e.process(batches[r])
    engs.append(e)
words = engs[0].slab_words(halo)
buf = torch.zeros(words, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()   # (the fill runs on torch's stream, the slab kernels on the handle's own non-blocking one)
engs[0].slab_export(0, plans[0]["end"], halo, buf.data_ptr())
engs[0].slab_clear(0, plans[0]["end"], halo)
crossed = int((buf != 0).sum())
engs[1].slab_add(0, plans[1]["begin"], halo, buf.data_ptr())
got = []
for e in engs:
    rows = e.finalize()
    got += list(zip(rows["pos"].tolist(), rows["strand"].tolist(), rows["n_called"].tolist(), rows["n_mod"].tolist()))
    e.close()
orc = O.Oracle([("m", "CG")], [0.8], ["chrS"])
orc.add_contig("chrS", ref)
for b in batches:
    orc.process(b)
w = orc.rows()
want = sorted(zip(w["pos"].tolist(), w["strand"].tolist(), w["n_called"].tolist(), w["n_mod"].tolist()))
print(json.dumps({"crossed": crossed, "rows": len(want), "equal": sorted(got) == want}))
'''


@pytest.mark.parametrize("stream_mode", [2, 1], ids=["stream", "tiles"])
def test_interval_sharding_with_halo_slabs_on_device(stream_mode):
    """Two handles own neighbouring intervals of one contig (as two ranks would); the left one's halo slab is exported,
    cleared, and added into the right one's planes with the library's slab kernels; the union equals the unsharded
    oracle bit for bit.  (Calls past a shard's planes go to the side table from either kernel.)"""
    r = subprocess.run([sys.executable, "-c", SHARD_WORKER % (ROOT, stream_mode)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert res["crossed"] > 0 and res["rows"] > 1000 and res["equal"], res


COALESCE_WORKER = r'''
import json, sys
sys.path.insert(0, %r)
STREAM_MODE = %d
import numpy as np
import torch
torch.zeros(1, device="cuda")
import minimod_amd
from minimod_amd import synth
from oracle import oracle as O
ref = synth.reference(21, 4 << 20)
bs = [synth.batch(ref, i * 300, 300, seed=9, n_reads_total=2100, with_order=False) for i in range(7)]
whole = synth.concat(bs)
dev = {k: torch.from_numpy(whole[k].view(np.uint8).reshape(-1)).cuda() for k in ("reads", "cigar", "seq", "mm", "ml")}
def window(i):
    return dict(reads=dev["reads"].data_ptr() + 64 * 300 * i, cigar=dev["cigar"].data_ptr(), seq=dev["seq"].data_ptr(), mm=dev["mm"].data_ptr(),
                ml=dev["ml"].data_ptr(), n_reads=300, n_cigar_words=len(whole["cigar"]), n_seq_bytes=len(whole["seq"]), n_mm_bytes=len(whole["mm"]),
                n_ml_bytes=len(whole["ml"]), max_n_cigar=int(bs[i]["max_n_cigar"]), max_l_qseq=int(bs[i]["max_l_qseq"]))
orc = O.Oracle([("m", "CG"), ("h", "CG")], [0.8, 0.7], ["chrS"]); orc.add_contig("chrS", ref)
for b in bs: orc.process(b, threads=8)
want = orc.rows()
key = lambda r, io: list(zip(r["pos"].tolist(), r["strand"].tolist(), r["code"].tolist(), r["n_called"].tolist(), r["n_mod"].tolist()))
out = {}
for name, coalesce, order in (("off", 1, range(7)), ("groups_of_3", 3, range(7)), ("one_group", 16, range(7)), ("broken_runs", 4, [0, 1, 3, 4, 5, 2, 6])):
    eng = minimod_amd.FreqEngine([("m", "CG", 0.8), ("h", "CG", 0.7)], [("chrS", len(ref), ref)], coalesce=coalesce, stream_mode=STREAM_MODE)
    tickets = [eng.submit_device(window(i)) for i in order]
    sizes = {}
    for t in tickets: sizes[t] = eng.ticket_batches(t)
    if name == "groups_of_3":
        eng.wait(tickets[-1])          # the open group of one is launched by the wait
    got = eng.finalize(); eng.close()   # finalize launches whatever is still gathered
    out[name] = {"equal": key(got, None) == key(want, None), "tickets": tickets, "sizes": [sizes[t] for t in dict.fromkeys(tickets)]}
# a failing read inside a gathered group: reported with its index counted from the group's first read
bad = [dict(b) for b in bs[:3]]
whole2 = synth.concat(bad)
cg = whole2["cigar"].copy(); r = whole2["reads"][450]; cg[int(r["cigar_off"])] = (5 << 4) | 5   # a hard clip in read 150 of window 1
dev2 = {k: torch.from_numpy((cg if k == "cigar" else whole2[k]).view(np.uint8).reshape(-1)).cuda() for k in ("reads", "cigar", "seq", "mm", "ml")}
def window2(i):
    return dict(reads=dev2["reads"].data_ptr() + 64 * 300 * i, cigar=dev2["cigar"].data_ptr(), seq=dev2["seq"].data_ptr(), mm=dev2["mm"].data_ptr(),
                ml=dev2["ml"].data_ptr(), n_reads=300, n_cigar_words=len(cg), n_seq_bytes=len(whole2["seq"]), n_mm_bytes=len(whole2["mm"]),
                n_ml_bytes=len(whole2["ml"]), max_n_cigar=int(bs[i]["max_n_cigar"]), max_l_qseq=int(bs[i]["max_l_qseq"]))
eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)], coalesce=4, stream_mode=STREAM_MODE)
ts = [eng.submit_device(window2(i)) for i in range(3)]
try:
    eng.wait(ts[0]); out["error"] = None
except minimod_amd.engine.MinimodHipError as e:
    out["error"] = [e.code, e.read]
eng.close()
print(json.dumps(out))
'''


@pytest.mark.parametrize("stream_mode", [2, 1], ids=["stream", "tiles"])
def test_coalesced_windows_of_a_resident_read_set(stream_mode):
    """mm_freq_opts_t.coalesce: consecutive -K windows of one resident read set share a launch; same rows as the oracle for
    every grouping, tickets shared inside a group, a submit that does not continue the group starts a new one, and a read
    error is reported relative to the group's first read (the stream kernel hands the read with the hard clip to the tile
    pipeline, which names it)."""
    r = subprocess.run([sys.executable, "-c", COALESCE_WORKER % (ROOT, stream_mode)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    for name in ("off", "groups_of_3", "one_group", "broken_runs"):
        assert res[name]["equal"], (name, res[name])
    assert res["off"]["sizes"] == [1, 1, 1, 1] or len(set(res["off"]["tickets"])) == 4      # four slots, every submit its own launch
    assert res["groups_of_3"]["tickets"][0] == res["groups_of_3"]["tickets"][2] != res["groups_of_3"]["tickets"][3]
    assert res["groups_of_3"]["sizes"] == [3, 3, 1]
    assert res["one_group"]["sizes"] == [7]
    assert res["broken_runs"]["sizes"] == [2, 3, 1, 1]
    assert res["error"] == [1, 450]


FALLBACK_WORKER = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np
import torch
torch.zeros(1, device="cuda")
import minimod_amd
from oracle import oracle as O
from oracle import pybam
from tests.cases import KAT_REF, KAT_SEQ, kat_records, kat2_records
# reads the tile pipeline hands to the fused kernel (five code letters in one group; groups on different bases) between
# ordinary ones, as two windows of one flattened read set gathered into one launch
recs = kat_records() + [
    pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+abcdm,0,0;", [1, 2, 3, 4, 250, 5, 6, 7, 8, 9]),
    pybam.make_record(0, 2, 0, KAT_SEQ, "20M", "C+m?,0;G+m?,1;", [255, 200]),
] + kat2_records() + kat_records()
b = pybam.flatten(recs)
dev = {k: torch.from_numpy(b[k].view(np.uint8).reshape(-1)).cuda() for k in ("reads", "cigar", "seq", "mm", "ml")}
n = len(recs)
def window(lo, hi):
    return dict(reads=dev["reads"].data_ptr() + 64 * lo, cigar=dev["cigar"].data_ptr(), seq=dev["seq"].data_ptr(), mm=dev["mm"].data_ptr(), ml=dev["ml"].data_ptr(),
                n_reads=hi - lo, n_cigar_words=len(b["cigar"]), n_seq_bytes=len(b["seq"]), n_mm_bytes=len(b["mm"]), n_ml_bytes=len(b["ml"]),
                max_n_cigar=int(b["reads"]["n_cigar"].max()), max_l_qseq=int(b["reads"]["l_qseq"].max()))
out = {}
for name, kw in (("m", {}), ("m_ins_hap", dict(insertions=True, haplotypes=True))):
    orc = O.Oracle([("m", "CG")], [0.8], ["chrT"], **kw); orc.add_contig("chrT", KAT_REF.encode()); orc.process(b)
    want = orc.rows()
    eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrT", len(KAT_REF), KAT_REF.encode())], coalesce=4, **kw)
    t = [eng.submit_device(window(0, 5)), eng.submit_device(window(5, 9)), eng.submit_device(window(9, n))]
    sizes = eng.ticket_batches(t[0])
    got = eng.finalize(); eng.close()
    key = lambda r, io: sorted(zip(r["pos"].tolist(), r["strand"].tolist(), r["code"].tolist(), r[io].tolist(), r["hp"].tolist(), r["n_called"].tolist(), r["n_mod"].tolist()))
    out[name] = {"equal": key(got, "ins_offset") == key(want, "ins_off"), "rows": int(len(want)), "one_ticket": len(set(t)) == 1, "members": sizes}
print(json.dumps(out))
'''


def test_gathered_windows_with_reads_for_the_fused_kernel():
    """A gathered launch whose windows hold reads the tile pipeline does not cover (the fallback list is worked off when the
    group's ticket is waited for or the counters are needed): same rows as the oracle."""
    r = subprocess.run([sys.executable, "-c", FALLBACK_WORKER % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    for name in ("m", "m_ins_hap"):
        assert res[name]["equal"] and res[name]["rows"] > 5 and res[name]["one_ticket"] and res[name]["members"] == 3, res


ORDER_WORKER = r'''
import sys
sys.path.insert(0, %r)
import minimod_amd
from tests.cases import KAT_REF, kat_records
from oracle import pybam
eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrT", len(KAT_REF), KAT_REF.encode())])
eng.wait(eng.submit(pybam.flatten(kat_records())))
import torch
x = torch.arange(8, device="cuda").sum().item()
rows = eng.finalize(); eng.close()
hip = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
print(x, len(rows["pos"]), len(hip))
'''


def test_library_first_then_torch_share_one_hip_runtime():
    """The ctypes loader and torch must end up on one copy of libamdhip64 whichever comes first:
    create an engine and run a batch before torch is imported, then use the GPU from torch."""
    r = subprocess.run([sys.executable, "-c", ORDER_WORKER % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    x, n_rows, n_hip = r.stdout.decode().strip().splitlines()[-1].split()
    assert int(x) == 28 and int(n_rows) > 0 and int(n_hip) == 1


def test_bench_e2e_devices_leg():
    """`bench.py --e2e-gbases G --e2e-devices 0,0`: the steady-state end-to-end job (one BAM written in pieces, its index merged from the
    pieces') through `minimod freq` once as a single run and once as two --devices workers sharing the box's GPU -- same bytes, and
    the line says what every worker spent where."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--e2e-gbases", "0.06", "--e2e-devices", "0,0", "--batch", "1024"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["parity_vs_cpu"]["byte_identical"] and d["reads"] > 3000
    dv = d["devices_run"]
    assert dv["n_workers"] == 2 and dv["byte_identical_to_single_run"] and len(dv["workers"]) == 2
    assert sum(w["reads"] for w in dv["workers"]) == d["reads"] and all(w["device"] == 0 for w in dv["workers"])
    assert dv["parent_merge_s"] is not None and dv["wall_s"] > 0
