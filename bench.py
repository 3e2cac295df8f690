#!/usr/bin/env python3
"""bench.py -- `minimod freq` hot path on MI355X: Mbases/s on synthetic ONT-shape reads.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run with one
rank per GPU.  Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1], "C2"): 100k ONT-shape reads (~15 kb) on one ~50 Mb reference interval per GPU,
5mC `-c m[CG] -m 0.8`, -K 4096.  A "step" is one -K 4096 batch through the hot path (kernel K1) with the batch
already resident in HBM; steps cycle over the 25 resident batches of the rank's shard.  N>1 is WEAK scaling: every rank
owns its own 50 Mb interval of one long contig with its own 100k reads (reads routed by start position, SURVEY.md
section 8e).  The timed region is exactly the K steps (barrier + synchronize on both sides, max over ranks).  The path's
only exchange, one halo-slab send/recv to the right neighbour (RCCL), happens once per job after the last batch, not per
step: it runs right behind the timed steps, is timed on its own and reported as `final_reduce` (with the throughput
these K steps would give as a whole job, `value_incl`).

`roofline.achieved` = algorithmic bytes per batch / mean device time of the batch's hot-path launches (k_scan_reads,
k_sum_tiles, k_call_tiles, fallback) from HIP events recorded by the library on the launch stream (the sum of the
four kernels, not just the largest one: the conservative reading); algorithmic bytes follow SURVEY.md section 8(d):
    B_read = 40 + 4*n_cigar + ceil(l_qseq/2) + |MM| + |ML| + 2*n_lookups + 16*n_updates
with n_lookups / n_updates tallied by the kernel itself in an untimed pass.
`cpu_baseline` = the oracle (oracle/freq_oracle.c, a bit-exact CPU restatement of the reference's algorithm: the
reference binary itself needs htslib, which this image lacks) on a bounded sample of the same batches, kind "port".
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INTERVAL = 48 << 20          # 50,331,648 positions per GPU ("one 50 Mb contig")
HALO = 1 << 18               # counters kept past the right edge of an interval
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU")
    ap.add_argument("--batch", type=int, default=4096, help="-K")
    ap.add_argument("--seed", type=int, default=0x5EED)
    ap.add_argument("--cpu-sample-batches", type=int, default=0, help="0 = choose for ~15 s of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump", default="", help="write the device results of rank 0's first batches (freq: rows of the first two "
                                                 "batches; view: rows of the first batch) to this .npz file; tests/ compare it with the oracle")
    ap.add_argument("--max-len", type=float, default=0.0, help="experiment: cap read length (0 = 200 kb)")
    ap.add_argument("--streams", type=int, default=1, help="1: every launch on one explicit stream; >1: the library's per-slot streams (up to 4 batches overlap)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra overlapped-streams measurement (use when profiling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU testing of the N>1 path)")
    ap.add_argument("--natural-order", action="store_true", help="do not process longest reads first")
    ap.add_argument("--mode", default="freq", choices=["freq", "view"],
                    help="freq = the headline metric (default); view = the same batches through `minimod view` (SURVEY.md 8f row 1), "
                         "rows ordered and left in HBM; an extra measurement, not the driver's contract line")
    return ap.parse_args()


def shard_plan(rank, world, interval=INTERVAL, halo=HALO):
    """Owned interval, halo and contig length for a rank (pure function, covered by the CPU tests).  Reads are routed
    by alignment START to the owner of that position; their calls may run past the owner's right edge by at most one
    read's reference span, which the halo must cover."""
    contig_len = world * interval + halo
    begin = rank * interval
    end = (rank + 1) * interval if rank < world - 1 else contig_len
    return {"contig_len": contig_len, "begin": begin, "end": end, "halo": halo if rank < world - 1 else 0,
            "read_begin": begin, "read_len": interval}


def exchange_halos(rank, world, export_fn, add_fn, make_buf, dist):
    """Send my halo slab to rank+1, add the slab received from rank-1.  export_fn(buf) fills buf with my halo slab,
    add_fn(buf) adds a received slab into my planes; make_buf() allocates a slab tensor.  Works with any
    torch.distributed backend (RCCL on GPUs, gloo in the CPU tests)."""
    if world == 1:
        return
    ops = []
    send = recv = None
    if rank < world - 1:
        send = make_buf()
        export_fn(send)
        ops.append(dist.P2POp(dist.isend, send, rank + 1))
    if rank > 0:
        recv = make_buf()
        ops.append(dist.P2POp(dist.irecv, recv, rank - 1))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if recv is not None:
        add_fn(recv)


def gen_reference(plan, seed):
    """A rank's reference: its own interval (+ halo + one read span past it) is generated, the rest stays 'N'."""
    from minimod_amd import synth
    ref = np.full(plan["contig_len"], ord("N"), dtype=np.uint8)
    g_end = min(plan["contig_len"], plan["end"] + plan["halo"] + (1 << 20))
    ref[plan["begin"]:g_end] = synth.reference_slice(seed, plan["begin"], g_end - plan["begin"])
    return ref


def gen_batch(ref, plan, rank, seed, reads, batch, bi, max_len=0.0):
    """Batch `bi` of a rank's synthetic reads (deterministic: tests regenerate it to check a --dump file)."""
    from minimod_amd import synth
    first = bi * batch
    n = min(batch, reads - first)
    return synth.batch(ref, first, n, seed=seed + 7919 * rank, contig_len=plan["contig_len"], n_reads_total=reads,
                       region_begin=plan["read_begin"], region_len=plan["read_len"], max_len=max_len)


def algorithmic_bytes(reads, lookups, updates):
    """bytes the path must touch, summed over a batch."""
    n = len(reads)
    return int(40 * n + 4 * int(reads["n_cigar"].sum()) + int(((reads["l_qseq"].astype(np.int64) + 1) // 2).sum()) +
               int(reads["mm_len"].sum()) + int(reads["ml_len"].sum()) + 2 * lookups + 16 * updates)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)   # testing: several ranks may share one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    host_staged = world > 1 and args.backend != "nccl"   # gloo moves CPU tensors

    import minimod_amd
    from minimod_amd import engine, synth

    plan = shard_plan(rank, world)
    t0 = time.time()
    ref = gen_reference(plan, args.seed)
    n_batches = (args.reads + args.batch - 1) // args.batch

    def gen(bi):
        return gen_batch(ref, plan, rank, args.seed, args.reads, args.batch, bi, args.max_len)

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        host_batches = list(ex.map(gen, range(n_batches)))
    t_gen = time.time() - t0

    if args.mode == "view":
        eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", plan["contig_len"], ref)], device=local_rank, view=True)
    else:
        eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", plan["contig_len"], ref)], device=local_rank,
                                     intervals=[(0, plan["begin"], plan["end"], plan["halo"])])
    # ---- make the batches resident in HBM (torch owns the memory: plumbing only)
    dev_batches = []
    keep = []
    for hb in host_batches:
        d = {}
        for k in ("reads", "cigar", "seq", "mm", "ml"):
            t = torch.from_numpy(hb[k].view(np.uint8).reshape(-1)).to(dev)
            keep.append(t)
            d[k] = t.data_ptr()
        if not args.natural_order:
            items = engine.plan_batch(hb["reads"])
            t = torch.from_numpy(items.view(np.uint8).reshape(-1)).to(dev)
            keep.append(t)
            d["order"] = t.data_ptr()
            d["n_order"] = len(items)
        d.update(n_reads=len(hb["reads"]), n_cigar_words=len(hb["cigar"]), n_seq_bytes=len(hb["seq"]),
                 n_mm_bytes=len(hb["mm"]), n_ml_bytes=len(hb["ml"]), max_n_cigar=hb["max_n_cigar"],
                 max_l_qseq=hb["max_l_qseq"])
        dev_batches.append(d)
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(device=dev)   # one explicit HIP stream carries every K1 launch
    stream = tstream.cuda_stream
    batch_bases = [hb["n_bases"] for hb in host_batches]

    if args.mode == "view":
        return bench_view(args, eng, host_batches, dev_batches, batch_bases, stream, rank, world, dist, dev, plan, ref, t_gen)

    # ---- untimed tally pass: lookups/updates per batch for the algorithmic-bytes figure
    eng.stats_enable(True)
    alg_bytes = []
    for hb, db in zip(host_batches, dev_batches):
        eng.wait(eng.submit_device(db, stream))
        st = eng.stats_get()
        alg_bytes.append(algorithmic_bytes(hb["reads"], st["lookups"], st["dense_updates"] + st["side_updates"]))
    eng.stats_enable(False)
    eng.reset()

    def run_steps(n, first_step=0, use_stream=stream):
        tickets, bases, kms, abytes = [], 0, [], 0
        for s in range(n):
            bi = (first_step + s) % n_batches
            t = eng.submit_device(dev_batches[bi], use_stream)
            tickets.append((t, bi))
            bases += batch_bases[bi]
            abytes += alg_bytes[bi]
            if len(tickets) >= 3:   # the library has 4 slots; keep the host a few launches ahead of the device
                tk, _ = tickets.pop(0)
                eng.wait(tk)
                kms.append(eng.kernel_ms(tk))
        for tk, _ in tickets:
            eng.wait(tk)
            kms.append(eng.kernel_ms(tk))
        return bases, kms, abytes

    # ---- warm-up, then the timed region: barrier + sync on both sides, max over ranks
    run_steps(args.warmup)
    eng.reset()
    slab_words = eng.slab_words(HALO)

    def make_buf():
        return torch.empty(slab_words, dtype=torch.int64, device="cpu" if host_staged else dev)

    def export_fn(buf):
        dbuf = torch.empty(slab_words, dtype=torch.int64, device=dev) if host_staged else buf
        eng.slab_export(0, plan["end"], plan["halo"], dbuf.data_ptr(), stream)
        eng.slab_clear(0, plan["end"], plan["halo"], stream)
        tstream.synchronize()          # the slab must be complete before RCCL (another stream) reads it
        if host_staged:
            buf.copy_(dbuf.cpu())

    def add_fn(buf):
        dbuf = buf.to(dev) if host_staged else buf
        torch.cuda.synchronize()       # the received slab is complete before our stream adds it
        eng.slab_add(0, plan["begin"], HALO, dbuf.data_ptr(), stream)
        tstream.synchronize()

    if world > 1:
        # warm the point-to-point path too (RCCL sets up its send/recv channels on first use: that must not land in
        # the timed region), then start from clean counters again
        exchange_halos(rank, world, export_fn, add_fn, make_buf, dist)
        eng.reset()
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bases, kms, abytes = run_steps(args.steps, first_step=args.warmup, use_stream=stream if args.streams <= 1 else None)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # The job's one exchange: each rank's halo slab goes to its right neighbour after the LAST batch (once per job, not per
    # step: a 30x genome is thousands of steps).  It runs here, right behind the K timed steps, and is timed on its own.
    t2 = time.perf_counter()
    exchange_halos(rank, world, export_fn, add_fn, make_buf, dist)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    reduce_s = time.perf_counter() - t2
    if world > 1:
        rdev = "cpu" if host_staged else dev
        tt = torch.tensor([elapsed, reduce_s], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, reduce_s = float(tt[0].item()), float(tt[1].item())
        tb = torch.tensor([bases], dtype=torch.int64, device=rdev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases = int(tb.item())
    else:
        total_bases = bases

    # extra, outside the contract's timed region: the same steps with the library's per-slot streams, i.e. up to three
    # batches in flight as the CLI's load/process overlap gives (kernels of consecutive batches overlap on the device)
    overlap = None
    if world == 1 and args.streams <= 1 and not args.no_extra:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ob, _, _ = run_steps(args.steps, first_step=args.warmup, use_stream=None)
        torch.cuda.synchronize()
        overlap = {"value": ob / (time.perf_counter() - t1) / 1e6, "unit": "Mbases/s",
                   "note": "same steps on the library's per-slot streams (3 batches in flight); not the contract's timed region"}

    result = None
    if rank == 0:
        mean_ms = float(np.mean(kms))
        achieved = (abytes / len(kms)) / (mean_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_k1.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "minimod freq Mbases/sec", "value": total_bases / elapsed / 1e6, "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "C2: %d ONT-shape reads (~15 kb) per GPU on a %.1f Mb interval, -c m[CG] -m 0.8, -K %d, "
                                   "batches resident in HBM" % (args.reads, INTERVAL / 1e6, args.batch),
                       "reads_per_gpu": args.reads, "batch_reads": args.batch, "mean_read_len": int(np.mean(
                           np.concatenate([hb["reads"]["l_qseq"] for hb in host_batches]))),
                       "sharding": "interval per GPU + halo slab to the right neighbour" if world > 1 else "single GPU",
                       "read_order": "natural, unsplit" if args.natural_order else "mm_freq_plan_batch (long reads split, costliest first)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "freq hot path per batch = k_scan_reads + k_sum_tiles + k_call_tiles (+ k_freq_reads on the fallback list), HIP events around the four launches",
                         "kernel_ms_mean": mean_ms, "algorithmic_bytes_per_launch": abytes / len(kms),
                         "bytes_per_base": abytes / max(bases, 1)},
            "gen_seconds": t_gen,
        }
        if world > 1:
            result["final_reduce"] = {"ms": reduce_s * 1e3, "value_incl": total_bases / (elapsed + reduce_s) / 1e6, "unit": "Mbases/s",
                                      "note": "halo slabs (%d positions x planes x 8 B) to the right neighbour, once per job after the last "
                                              "step; timed on its own, max over ranks; value_incl = throughput if these K steps were the "
                                              "whole job" % HALO}
        if overlap:
            result["overlapped_streams"] = overlap
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, host_batches, plan, ref)
        if args.dump:
            chk = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", plan["contig_len"], ref)], device=local_rank)
            for hb in host_batches[:2]:
                chk.process(hb)
            np.savez(args.dump, rows=chk.finalize())
            chk.close()
            result["dump"] = {"file": args.dump, "batches": min(2, len(host_batches))}
        print(json.dumps(result))
        sys.stdout.flush()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def bench_view(args, eng, host_batches, dev_batches, batch_bases, stream, rank, world, dist, dev, plan, ref, t_gen):
    """`minimod view` on the same resident batches: a step = call kernels + device-side ordering of the batch's rows
    (counting sort by read, one small sort per read; view_kernels.hip.h), rows left in HBM (mm_view_fetch_device).  Reads shard by interval
    exactly as for freq; rows are per read, so there is no exchange at all."""
    import torch
    n_batches = len(dev_batches)
    # untimed tally pass: reference-word lookups per batch (kernel tallies) and rows per batch
    eng.stats_enable(True)
    alg_bytes, rows_per_batch = [], []
    for hb, db in zip(host_batches, dev_batches):
        _, n = eng.fetch_view(eng.submit_device(db, stream), device=True)
        st = eng.stats_get()
        rows_per_batch.append(n)
        alg_bytes.append(algorithmic_bytes(hb["reads"], st["lookups"], 0) + 16 * n)
    eng.stats_enable(False)

    def run_steps(n, first_step=0):
        tickets, bases, kms, abytes, rows = [], 0, [], 0, 0
        for s in range(n):
            bi = (first_step + s) % n_batches
            tickets.append(eng.submit_device(dev_batches[bi], stream))
            bases += batch_bases[bi]
            abytes += alg_bytes[bi]
            if len(tickets) >= 3:
                tk = tickets.pop(0)
                rows += eng.fetch_view(tk, device=True)[1]
                kms.append(eng.kernel_ms(tk))
        for tk in tickets:
            rows += eng.fetch_view(tk, device=True)[1]
            kms.append(eng.kernel_ms(tk))
        return bases, kms, abytes, rows

    run_steps(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bases, kms, abytes, rows = run_steps(args.steps, first_step=args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    total_bases = bases
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tb = torch.tensor([bases], dtype=torch.int64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases = int(tb.item())
    result = None
    if rank == 0:
        mean_ms = float(np.mean(kms))
        step_ms = elapsed / args.steps * 1e3
        result = {
            "metric": "minimod view Mbases/sec", "value": total_bases / elapsed / 1e6, "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "C2 batches through view: %d ONT-shape reads per GPU, -c m[CG], -K %d, batches resident in HBM, "
                                   "ordered rows left in HBM" % (args.reads, args.batch),
                       "rows_per_step": rows / args.steps, "sharding": "interval per GPU, no exchange" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": (abytes / args.steps) / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (abytes / args.steps) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "whole step = k_scan_reads + k_sum_tiles + k_call_tiles<view> + k_view_offsets + k_view_scatter + "
                                   "k_view_sort + k_view_sort_big, wall clock per step; HIP events around the same launches: kernel_ms_mean",
                         "kernel_ms_mean": mean_ms, "algorithmic_bytes_per_launch": abytes / args.steps},
            "gen_seconds": t_gen,
        }
        if args.dump:
            np.savez(args.dump, rows=eng.view(host_batches[0]))
            result["dump"] = {"file": args.dump, "batches": 1}
        if not args.no_cpu_baseline:
            from oracle import oracle as O
            cores = os.cpu_count() or 1
            orc = O.Oracle([("m", "CG")], [0.8], ["chrS"])
            orc.set_view(True)
            orc.add_contig("chrS", ref)
            tc = time.perf_counter()
            orc.process(host_batches[0], threads=cores)
            tc = time.perf_counter() - tc
            orc.close()
            result["cpu_baseline"] = {"value": batch_bases[0] / tc / 1e6, "unit": "Mbases/s", "cores": cores, "kind": "port",
                                      "sample": "first -K %d batch (%d bases), oracle view mode with %d threads, process step only"
                                                % (args.batch, batch_bases[0], cores)}
        print(json.dumps(result))
        sys.stdout.flush()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def cpu_baseline(args, host_batches, plan, ref):
    """The oracle (bit-exact CPU restatement) timed on this host's cores over a bounded sample of the same batches."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    orc = O.Oracle([("m", "CG")], [0.8], ["chrS"])
    orc.add_contig("chrS", ref)
    n = args.cpu_sample_batches
    t0 = time.perf_counter()
    orc.process(host_batches[0], threads=cores)
    t1 = time.perf_counter() - t0
    bases = host_batches[0]["n_bases"]
    total_t = t1
    if n == 0:
        n = int(max(1, min(len(host_batches), round(15.0 / max(t1, 1e-3)))))
    for hb in host_batches[1:n]:
        t0 = time.perf_counter()
        orc.process(hb, threads=cores)
        total_t += time.perf_counter() - t0
        bases += hb["n_bases"]
    orc.close()
    return {"value": bases / total_t / 1e6, "unit": "Mbases/s", "cores": cores, "kind": "port",
            "sample": "first %d of %d -K %d batches (%d bases), oracle/freq_oracle.c with %d threads, process step only"
                      % (min(n, len(host_batches)), len(host_batches), args.batch, bases, cores)}


if __name__ == "__main__":
    main()
