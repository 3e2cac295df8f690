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
