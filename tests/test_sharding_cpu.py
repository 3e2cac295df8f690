"""Multi-GPU path on CPU: world_size-2 gloo run of the interval sharding + halo-slab exchange that bench.py uses over
RCCL.  Each rank counts its own reads with the oracle into a dense slab covering its interval + halo, the halo goes to
the right neighbour through bench.exchange_halos, and the union must equal the unsharded result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTERVAL, HALO = 1 << 20, 1 << 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reads_for(rank, plan, ref):
    from minimod_amd import synth
    return synth.batch(ref, 0, 900, seed=11 + 7919 * rank, contig_len=plan["contig_len"], n_reads_total=900,
                       region_begin=plan["read_begin"], region_len=plan["read_len"], median_len=3000.0, max_len=30000.0)


def _oracle_rows(batches, ref):
    from oracle import oracle as O
    o = O.Oracle([("m", "CG")], [0.8], ["chrS"])
    o.add_contig("chrS", ref)
    for b in batches:
        o.process(b)
    return o.rows()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    from minimod_amd import synth
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    plan = bench.shard_plan(rank, world, INTERVAL, HALO)
    ref = synth.reference(5, plan["contig_len"])
    rows = _oracle_rows([_reads_for(rank, plan, ref)], ref)
    seg_len = plan["end"] + plan["halo"] - plan["begin"]
    planes = np.zeros((2, seg_len), dtype=np.uint64)          # [strand][pos - begin] = n_called | n_mod << 32
    assert rows["pos"].min() >= plan["begin"] and rows["pos"].max() < plan["begin"] + seg_len, "halo too small"
    planes[rows["strand"], rows["pos"] - plan["begin"]] = rows["n_called"].astype(np.uint64) | (rows["n_mod"].astype(np.uint64) << np.uint64(32))

    def make_buf():
        return torch.zeros(2 * HALO, dtype=torch.int64)

    def export_fn(buf):
        off = plan["end"] - plan["begin"]
        buf.copy_(torch.from_numpy(planes[:, off:off + HALO].reshape(-1).view(np.int64).copy()))
        planes[:, off:off + HALO] = 0

    def add_fn(buf):
        v = buf.numpy().view(np.uint64).reshape(2, HALO)
        lo = (planes[:, :HALO] & np.uint64(0xFFFFFFFF)) + (v & np.uint64(0xFFFFFFFF))
        hi = (planes[:, :HALO] >> np.uint64(32)) + (v >> np.uint64(32))
        planes[:, :HALO] = lo | (hi << np.uint64(32))

    bench.exchange_halos(rank, world, export_fn, add_fn, make_buf, dist)
    owned = planes[:, :plan["end"] - plan["begin"]]
    s, p = np.nonzero(owned)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), strand=s, pos=p + plan["begin"],
             n_called=(owned[s, p] & np.uint64(0xFFFFFFFF)), n_mod=(owned[s, p] >> np.uint64(32)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_plan_covers_the_contig():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8):
        plans = [bench.shard_plan(r, world) for r in range(world)]
        assert plans[0]["begin"] == 0 and plans[-1]["end"] == plans[-1]["contig_len"]
        for a, b in zip(plans, plans[1:]):
            assert a["end"] == b["begin"] and a["halo"] == bench.HALO and a["read_begin"] + a["read_len"] == a["end"]
        assert plans[-1]["halo"] == 0 and plans[0]["begin"] % (1 << 20) == 0


@pytest.mark.timeout(300)
def test_world2_interval_sharding_equals_unsharded(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import bench
    from minimod_amd import synth
    plans = [bench.shard_plan(r, world, INTERVAL, HALO) for r in range(world)]
    ref = synth.reference(5, plans[0]["contig_len"])
    want = _oracle_rows([_reads_for(r, plans[r], ref) for r in range(world)], ref)
    got = []
    for r in range(world):
        z = np.load(str(tmp_path / ("rank%d.npz" % r)))
        got += list(zip(z["pos"].tolist(), z["strand"].tolist(), z["n_called"].tolist(), z["n_mod"].tolist()))
    want_l = sorted(zip(want["pos"].tolist(), want["strand"].tolist(), want["n_called"].tolist(), want["n_mod"].tolist()))
    assert len(want_l) > 1000
    # some of rank 0's own calls must really land past its right edge, or the test proves nothing
    own0 = _oracle_rows([_reads_for(0, plans[0], ref)], ref)
    assert int((own0["pos"] >= plans[0]["end"]).sum()) > 0
    assert sorted(got) == want_l


# ---- N > 1 as bench.py runs it by default: a genome of several contigs cut into one contiguous share per rank
GENOME = [("chrA", 1 << 20), ("chrB", 2 << 20), ("chrC", 1 << 20)]


def _genome_reads(rank, plan, refs):
    import bench
    return bench.plan_reads(plan, refs, rank, 77, 700, max_len=30000.0, median_len=3000.0)


def _genome_oracle(batches, refs):
    from oracle import oracle as O
    o = O.Oracle([("m", "CG")], [0.8], [n for n, _ in GENOME])
    for (n, _), r in zip(GENOME, refs):
        if r is not None:
            o.add_contig(n, r)
    for b in batches:
        o.process(b)
    return o.rows()


def _genome_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    plan = bench.genome_plan(rank, world, GENOME, HALO)
    refs = bench.plan_references(plan, 5)
    rows = _genome_oracle([_genome_reads(rank, plan, refs)], refs)
    key = lambda r: (int(r["tid"]), int(r["pos"]), int(r["strand"]))
    counts = {key(r): [int(r["n_called"]), int(r["n_mod"])] for r in rows}
    slab_len = max([x[2] for x in (plan["send"], plan["recv"]) if x] + [0])

    def make_buf():
        return torch.zeros(2 * 2 * slab_len, dtype=torch.int64)

    def export_fn(buf):                      # the counters past my right edge leave with the slab
        tid, pos, ln = plan["send"]
        a = buf.numpy().reshape(2, 2, slab_len)
        for k in [k for k in counts if k[0] == tid and pos <= k[1] < pos + ln]:
            a[k[2], :, k[1] - pos] = counts.pop(k)

    def add_fn(buf):
        tid, pos, ln = plan["recv"]
        a = buf.numpy().reshape(2, 2, slab_len)
        for strand, off in zip(*np.nonzero(a[:, 0, :])):
            c = counts.setdefault((tid, pos + int(off), int(strand)), [0, 0])
            c[0] += int(a[strand, 0, off]); c[1] += int(a[strand, 1, off])

    bench.exchange_slabs(rank, world, plan["send"] is not None, plan["recv"] is not None, export_fn, add_fn, make_buf, dist)
    # what is left must lie inside my intervals
    for (tid, pos, _s) in counts:
        assert any(iv["tid"] == tid and iv["begin"] <= pos < iv["end"] for iv in plan["intervals"]), (rank, tid, pos)
    np.save(os.path.join(out_dir, "g%d.npy" % rank), np.array([k + tuple(v) for k, v in counts.items()], dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


def test_genome_plan_partitions_the_contigs():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 3, 4, 8):
        for contigs in (GENOME, bench.genome_layout(world, 8 << 20)):
            plans = [bench.genome_plan(r, world, contigs) for r in range(world)]
            covered = {}
            for p in plans:
                for iv in p["intervals"]:
                    assert iv["begin"] % bench.CUT_ALIGN == 0 and iv["begin"] < iv["end"] <= contigs[iv["tid"]][1]
                    covered.setdefault(iv["tid"], []).append((iv["begin"], iv["end"]))
            for tid, (_n, l) in enumerate(contigs):       # every contig tiled exactly once, in order
                ivs = sorted(covered[tid])
                assert ivs[0][0] == 0 and ivs[-1][1] == l and all(a[1] == b[0] for a, b in zip(ivs, ivs[1:]))
            for a, b in zip(plans, plans[1:]):            # a cut inside a contig: left sends what right receives
                assert (a["send"] is None) == (b["recv"] is None)
                if a["send"]:
                    assert a["send"] == b["recv"] and a["intervals"][-1]["halo"] == a["send"][2] > 0
            assert plans[0]["recv"] is None and plans[-1]["send"] is None


@pytest.mark.timeout(300)
def test_world2_genome_sharding_equals_unsharded(tmp_path):
    """Two ranks own contiguous shares of a three-contig genome; the cut falls inside chrB, so rank 0's calls past it travel
    to rank 1 as a halo slab (bench.exchange_slabs over gloo); union == unsharded oracle."""
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.spawn(_genome_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import bench
    plans = [bench.genome_plan(r, world, GENOME, HALO) for r in range(world)]
    assert plans[0]["send"] == (1, 1 << 20, HALO) == plans[1]["recv"]
    refs_all = [bench.plan_references(p, 5) for p in plans]
    refs = [a if a is not None else b for a, b in zip(*refs_all)]
    for t in range(len(GENOME)):             # the two ranks' pieces of chrB are slices of one synthetic contig
        if refs_all[0][t] is not None and refs_all[1][t] is not None:
            refs[t] = np.where(refs_all[0][t] != ord("N"), refs_all[0][t], refs_all[1][t])
    want = _genome_oracle([_genome_reads(r, plans[r], refs_all[r]) for r in range(world)], refs)
    want_l = sorted(zip(want["tid"].tolist(), want["pos"].tolist(), want["strand"].tolist(), want["n_called"].tolist(), want["n_mod"].tolist()))
    got = []
    for r in range(world):
        got += [tuple(x) for x in np.load(str(tmp_path / ("g%d.npy" % r))).tolist()]
    own0 = _genome_oracle([_genome_reads(0, plans[0], refs_all[0])], refs_all[0])
    assert int(((own0["tid"] == 1) & (own0["pos"] >= (1 << 20))).sum()) > 0      # rank 0 really has calls past the cut
    assert len(want_l) > 1000 and sorted(got) == want_l


def test_bench_gpus_flag_launches_the_ranks_without_a_launcher():
    """`python bench.py --gpus N` with WORLD_SIZE unset starts N ranks itself (fresh child processes, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 set) and passes rank 0's line on; under an external launcher (WORLD_SIZE set) it does not."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MM_BENCH_LAUNCH_ONLY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d == {"launch_only": True, "n_gpus": 4, "rank": 0, "local_rank": 0, "master": "127.0.0.1"}
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env2)
    assert r.returncode == 0 and json.loads(r.stdout.decode().strip().splitlines()[-1])["n_gpus"] == 2


def test_config_c4_is_the_human_genome_at_eight_gpus():
    """bench.py --config C4 (BASELINE.json configs[3]): 370 MiB of reference and 775 000 reads per GPU; at eight GPUs the 24 contigs
    add up to a human genome and every rank's share is an eighth of it, cut at 64 kb-aligned positions."""
    import bench
    wl = bench.WORKLOADS["C4"]
    assert wl["reads"] == 775000 and wl["region"] == 370 << 20 and wl.get("genome")
    for world in (1, 2, 8):
        contigs = bench.genome_layout(world, wl["region"])
        assert len(contigs) == 24 and [n for n, _ in contigs][:3] == ["chr1", "chr2", "chr3"]
        total = sum(l for _, l in contigs)
        assert abs(total - world * wl["region"]) < 24 << 20
        if world == 8:
            assert 3.0e9 < total < 3.2e9
        owned = 0
        for r in range(world):
            plan = bench.genome_plan(r, world, contigs, bench.HALO)
            mine = sum(iv["end"] - iv["begin"] for iv in plan["intervals"])
            owned += mine
            assert abs(mine - total / world) <= bench.CUT_ALIGN
        assert owned == total
    what = wl["what"] % dict(reads=wl["reads"], mb=wl["region"] / 1e6, batch=4096)
    assert what.startswith("C4: 775000 ONT-shape reads") and "388.0 Mb" in what
