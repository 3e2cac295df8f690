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

At N=1 the line also carries the END-TO-END leg: the same reads written as a real BGZF BAM +
FASTA, `minimod_amd/bin/minimod freq -b -c m[CG] -m 0.8 -K 4096 -B 200M -t <cores>` run as a child process BEFORE this
process touches the GPU (`end_to_end`: wall, Mbases/s, the CLI's stage timers), and beside it the CPU path end to end
(`cpu_baseline_e2e`: oracle/freq_cpu_main.c = the oracle behind the same reader and formatter, `-t <cores>` on the same
file and `-t 1` on a bounded sample), with the two bedmethyl outputs compared byte for byte.

`--config C3` (HiFi-shape reads, `-c m[CG],h[CG] -m 0.8,0.7`) and `--config C5` (`--haplotypes --insertions`, 200x on a
5 Mb region) run the other BASELINE.json workloads through the same steps; extra measurements, not the driver's line.
"""
import hashlib
import re
import shutil
import subprocess
import tempfile
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
    ap.add_argument("--dump-timed", default="", help="write the rows the TIMED engine holds after the timed steps (its counters were reset "
                                                     "before the warm-up: with --warmup 0 and --steps = the number of batches they are one pass "
                                                     "over the workload at full size, through exactly the launches that were timed)")
    ap.add_argument("--dump", default="", help="write the device results of rank 0's first batches (freq: rows of the first two "
                                                 "batches; view: rows of the first batch) to this .npz file; tests/ compare it with the oracle")
    ap.add_argument("--max-len", type=float, default=0.0, help="experiment: cap read length (0 = 200 kb)")
    ap.add_argument("--streams", type=int, default=1, help="1: every launch on one explicit stream; >1: the library's per-slot streams (up to 4 batches overlap)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra overlapped-streams measurement (use when profiling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU testing of the N>1 path)")
    ap.add_argument("--host-plan", action="store_true", help="experiment: hand the library a host-made plan (mm_freq_plan_batch) "
                                                           "uploaded before the timed region instead of planning on the device inside it")
    ap.add_argument("--coalesce", type=int, default=32, help="mm_freq_opts_t.coalesce: consecutive -K windows of the resident read set that may share "
                                                             "one launch (1 = every step is its own launch)")
    ap.add_argument("--region-mb", type=float, default=0.0, help="experiment: reads per GPU spread over this many Mb instead of the workload's "
                                                                  "interval (depth = reads * 15 kb / region: counter contention at depth)")
    ap.add_argument("--force-fused", action="store_true", help="experiment: the fused one-wavefront-per-read kernel for every read")
    ap.add_argument("--no-stream", action="store_true", help="experiment: mm_freq_opts_t.stream_mode = 1 (every read through the tile pipeline)")
    ap.add_argument("--split-bases", type=int, default=0, help="experiment: part size of the device planning (0 = library default)")
    ap.add_argument("--single-contig", action="store_true", help="N > 1: one long contig cut into one interval per rank (round 1's layout) "
                                                               "instead of the 24-contig genome")
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5", "DOT"], help="BASELINE.json workload: C2 = the headline (default); DOT (not a BASELINE config): C2's reads "
                                                                                     "with the MM '.' flag (every unlisted C an implicit call)")
    ap.add_argument("--stream-slices", type=int, default=0, help="mm_freq_opts_t.stream_slices: 0 one position slice of a launch per XCD (default), 1 costliest first over the whole launch")
    ap.add_argument("--no-config-fracs", action="store_true", help="skip the short runs of the other BASELINE workloads (C3, C5, view) whose roofline fractions the default line carries as `config_fracs`")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end CLI leg and its CPU counterpart")
    ap.add_argument("--e2e-threads", type=int, default=0, help="-t of the end-to-end runs (0 = all host cores, at most 128)")
    ap.add_argument("--cpu-t1-batches", type=int, default=2, help="batches in the BAM the `-t 1` CPU run reads")
    ap.add_argument("--reps", type=int, default=11, help="repetitions of the timed region (each exactly --steps steps between barrier + synchronize): "
                                                          "`value` is the median repetition, `spread` has min / max")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-path leg (the same batches through mm_freq_submit, PCIe included) and the "
                                                                "one-launch-per-step leg")
    ap.add_argument("--e2e-gbases", type=float, default=0.0, help="opt-in, instead of the bench line: the STEADY-STATE end-to-end figure -- this many Gbases of "
                                                                    "reads (30x of a genome share: 12 = BASELINE configs[3] / 8) as one BGZF BAM, through the product "
                                                                    "CLI and through the CPU port, start-up included and excluded; prints its own JSON line")
    ap.add_argument("--e2e-devices", default="", help="with --e2e-gbases: also run the same job as `minimod freq --devices LIST` (one worker process per listed GPU, "
                    "e.g. 0,1,2,3 on a node, 0,0 to run two workers on the one GPU of a box): wall, every worker's loading / waiting / finalize time, the parent's merge and output "
                    "time, bytes compared with the single run's")
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


def exchange_slabs(rank, world, has_send, has_recv, export_fn, add_fn, make_buf, dist):
    """exchange_halos for a genome plan: only ranks whose share ends inside a contig send, only ranks whose share begins
    inside one receive (the neighbour relation is the same: rank -> rank + 1)."""
    if world == 1:
        return
    ops, recv = [], None
    if has_send:
        send = make_buf()
        export_fn(send)
        ops.append(dist.P2POp(dist.isend, send, rank + 1))
    if has_recv:
        recv = make_buf()
        ops.append(dist.P2POp(dist.irecv, recv, rank - 1))
    if ops:
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


# ---- N > 1: a genome of 24 contigs cut into contiguous shares (BASELINE.json configs[3])
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
        133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895,
        57227415]
GENOME_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
CUT_ALIGN = 1 << 16          # shares are cut at 64 kb-aligned positions


def genome_layout(world, per_gpu=INTERVAL):
    """24 contigs with hg38's length ratios and world * per_gpu positions in total (weak scaling: every GPU gets per_gpu
    positions of reference whatever the world size; at 8 GPUs and per_gpu = 388 Mb this is the human genome).  Lengths are
    multiples of 1 MiB (the synthetic reference comes in MiB chunks)."""
    mib = 1 << 20
    total = world * per_gpu
    lens = [max(mib, int(round(h / sum(HG38) * total / mib)) * mib) for h in HG38]
    return list(zip(GENOME_NAMES, lens))


def genome_plan(rank, world, contigs, halo=HALO, align=CUT_ALIGN):
    """Rank `rank`'s contiguous share of the concatenated genome (contigs in header order, cut points aligned): the
    intervals it owns, and the halo slab it sends to the right / receives from the left when a cut falls inside a contig.
    A read belongs to the owner of its start position (pure function, covered by the CPU tests)."""
    lens = [l for _, l in contigs]
    total = sum(lens)
    cut = lambda k: total if k >= world else (k * total // world) // align * align
    lo, hi = cut(rank), cut(rank + 1)
    intervals, off = [], 0
    for tid, l in enumerate(lens):
        b, e = max(lo, off), min(hi, off + l)
        if b < e:
            intervals.append({"tid": tid, "begin": b - off, "end": e - off, "halo": 0, "read_begin": b - off, "read_len": e - b})
        off += l
    send = recv = None
    if intervals:
        last, first = intervals[-1], intervals[0]
        if last["end"] < lens[last["tid"]]:          # the cut is inside a contig: counters past it are kept and sent on
            last["halo"] = min(halo, lens[last["tid"]] - last["end"])
            send = (last["tid"], last["end"], last["halo"])
        if first["begin"] > 0:
            recv = (first["tid"], first["begin"], min(halo, lens[first["tid"]] - first["begin"]))
    return {"contigs": list(contigs), "intervals": intervals, "send": send, "recv": recv, "share": hi - lo}


def single_contig_plan(rank, world, interval=INTERVAL, halo=HALO):
    """shard_plan() (one long contig cut into one interval per rank) in the form genome_plan() returns."""
    p = shard_plan(rank, world, interval, halo)
    iv = {"tid": 0, "begin": p["begin"], "end": p["end"], "halo": p["halo"], "read_begin": p["read_begin"], "read_len": p["read_len"]}
    return {"contigs": [("chrS", p["contig_len"])], "intervals": [iv], "send": (0, p["end"], p["halo"]) if p["halo"] else None,
            "recv": (0, p["begin"], halo) if rank > 0 else None, "share": interval, "single": p}


def plan_references(plan, seed):
    """The reference of every contig the rank owns a piece of: that piece (+ halo + one read span) generated, the rest 'N';
    contigs it owns nothing of stay None (not uploaded)."""
    from minimod_amd import synth
    refs = [None] * len(plan["contigs"])
    for iv in plan["intervals"]:
        tid, clen = iv["tid"], plan["contigs"][iv["tid"]][1]
        if refs[tid] is None:
            refs[tid] = np.full(clen, ord("N"), dtype=np.uint8)
        b = max(0, iv["begin"] - 64) // (1 << 20) * (1 << 20)   # (real bases in front of the piece too: a context match may begin there)
        e = min(clen, iv["end"] + iv["halo"] + (1 << 20))
        refs[tid][b:e] = synth.reference_slice(seed + 1000 * tid, b, e - b)
    return refs


def plan_reads(plan, refs, rank, seed, reads, max_len=0.0, **gen):
    """The rank's reads: every interval gets its share of `reads` by length, sorted by (contig, start) like a BAM."""
    from minimod_amd import synth
    share = sum(iv["read_len"] for iv in plan["intervals"])
    parts, left = [], reads
    for k, iv in enumerate(plan["intervals"]):
        n = left if k == len(plan["intervals"]) - 1 else min(left, int(round(reads * iv["read_len"] / share)))
        left -= n
        if n <= 0:
            continue
        clen = plan["contigs"][iv["tid"]][1]
        parts.append(synth.batch(refs[iv["tid"]], 0, n, seed=seed + 7919 * rank + 101 * k, contig_len=clen, n_reads_total=n, tid=iv["tid"],
                                 region_begin=iv["read_begin"], region_len=iv["read_len"], max_len=max_len, with_order=False, **gen))
    return synth.concat(parts)


# The BASELINE.json workloads this file can run.  `gen` = synthetic-read options, `mods` = -c / -m, `eng` = engine options.
WORKLOADS = {
    "C2": dict(gen=dict(), mods=[("m", "CG", 0.8)], eng=dict(), cli=["-c", "m[CG]", "-m", "0.8"],
               what="C2: %(reads)d ONT-shape reads (~15 kb) per GPU on a %(mb).1f Mb interval, -c m[CG] -m 0.8, -K %(batch)d, batches resident in HBM"),
    # BASELINE.json configs[3]: 30x of a human-sized genome over 8 GPUs = 370 MiB (388 M positions) of reference and 775 000 ONT-shape
    # reads (11.7 Gbases) PER GPU, 24 contigs with hg38's length ratios whatever N (at N = 8 the whole genome), contiguous shares cut at
    # 64 kb-aligned positions, halo slabs exchanged once behind the last step.  (The reads are made by the C generator, csrc/host/synth.c,
    # on the host's cores: ~20 s and ~10 GB of host memory per rank.)
    "C4": dict(gen=dict(), mods=[("m", "CG", 0.8)], eng=dict(), cli=["-c", "m[CG]", "-m", "0.8"], region=370 << 20, reads=775000, genome=True,
               what="C4: %(reads)d ONT-shape reads (~15 kb, 30x) per GPU on a %(mb).1f Mb share of a 24-contig genome with hg38's length ratios, "
                    "-c m[CG] -m 0.8, -K %(batch)d, batches resident in HBM"),
    "C3": dict(gen=dict(shape=1), mods=[("m", "CG", 0.8), ("h", "CG", 0.7)], eng=dict(), cli=["-c", "m[CG],h[CG]", "-m", "0.8,0.7"],
               what="C3: %(reads)d PacBio-HiFi-shape reads (~15 kb, MM '?' flag) per GPU on a %(mb).1f Mb interval, -c m[CG],h[CG] -m 0.8,0.7, "
                    "-K %(batch)d, batches resident in HBM"),
    "DOT": dict(gen=dict(dot_fraction=1.0), mods=[("m", "CG", 0.8)], eng=dict(), cli=["-c", "m[CG]", "-m", "0.8"],
                what="DOT: %(reads)d ONT-shape reads (~15 kb) with the MM '.' flag (all unlisted Cs are implicit calls) per GPU on a %(mb).1f Mb interval, "
                     "-c m[CG] -m 0.8, -K %(batch)d, batches resident in HBM"),
    "C5": dict(gen=dict(haplotypes=True, long_insertions=True), mods=[("m", "CG", 0.8)], eng=dict(insertions=True, haplotypes=True),
               cli=["-c", "m[CG]", "-m", "0.8", "--insertions", "--haplotypes"], region=5 << 20, reads=66000,
               what="C5: %(reads)d ONT-shape reads = 200x on a %(mb).1f Mb region, HP tags, CpG-carrying insertions, -c m[CG] -m 0.8 "
                    "--insertions --haplotypes, -K %(batch)d, batches resident in HBM"),
}


def gen_batch(ref, plan, rank, seed, reads, batch, bi, max_len=0.0, **gen):
    """Batch `bi` of a rank's synthetic reads (deterministic: tests regenerate it to check a --dump file)."""
    from minimod_amd import synth
    first = bi * batch
    n = min(batch, reads - first)
    return synth.batch(ref, first, n, seed=seed + 7919 * rank, contig_len=plan["contig_len"], n_reads_total=reads,
                       region_begin=plan["read_begin"], region_len=plan["read_len"], max_len=max_len, with_order=False, **gen)


def algorithmic_bytes(reads, lookups, updates):
    """bytes the path must touch, summed over a batch."""
    n = len(reads)
    return int(40 * n + 4 * int(reads["n_cigar"].sum()) + int(((reads["l_qseq"].astype(np.int64) + 1) // 2).sum()) +
               int(reads["mm_len"].sum()) + int(reads["ml_len"].sum()) + 2 * lookups + 16 * updates)


def _stage_timers(stderr_text):
    """The reference-format stage timers a CLI run prints at exit (src/freq_main.c:505-509) as a dict of seconds."""
    out = {}
    for key, pat in (("load", r"Data loading time: ([0-9.]+)"), ("process", r"Data processing time: ([0-9.]+)"),
                     ("merge", r"Data merging time: ([0-9.]+)"), ("sort", r"Data sorting time: ([0-9.]+)"),
                     ("output", r"Data output time: ([0-9.]+)"), ("reference", r"Reference genome loaded in ([0-9.]+)"),
                     ("reference", r"Reference loading time: ([0-9.]+)"), ("contexts", r"Reference contexts loaded in ([0-9.]+)"),
                     ("contexts", r"Reference contexts time: ([0-9.]+)"), ("gpu_runtime", r"GPU runtime ready [0-9.]+ sec after the process began \(waited ([0-9.]+)")):
        m = re.search(pat, stderr_text)
        if m:
            out[key] = float(m.group(1))
    return out



def usable_cores():
    """CPUs this process can actually use: the affinity mask, cut down to the cgroup's CPU quota if there is one (the GPU boxes
    show 256 hardware threads and grant 16 CPUs' worth of time: 128 workers there only preempt each other)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:   # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n

def run_end_to_end(args, wl, host_batches, contig, ref):
    """The END-TO-END leg: the workload's reads as a BGZF BAM + FASTA, through the product CLI (GPU) and through the CPU
    path (oracle behind the same reader and formatter), every run a child process.  Called before this process initialises
    the GPU.  Returns (end_to_end, cpu_baseline_e2e)."""
    from minimod_amd import synth
    from oracle import oracle as O
    cores = usable_cores()
    threads = args.e2e_threads if args.e2e_threads > 0 else min(cores, 128)
    cli = os.path.join(ROOT, "minimod_amd", "bin", "minimod")
    cpu_cli = O.build_cpu_cli()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (8 << 30) else None
    tmp = tempfile.mkdtemp(prefix="mm_e2e_", dir=base)
    try:
        t0 = time.perf_counter()
        bam, bam1, fa = os.path.join(tmp, "reads.bam"), os.path.join(tmp, "reads_t1.bam"), os.path.join(tmp, "ref.fa")
        contigs = [contig]
        synth.write_bam_parallel(bam, contigs, host_batches, filter_fodder=True, threads=min(32, cores))
        n1 = max(1, min(args.cpu_t1_batches, len(host_batches)))
        synth.write_bam_parallel(bam1, contigs, host_batches[:n1], filter_fodder=True, threads=min(32, cores))
        synth.write_fasta(fa, contig[0], ref)
        t_write = time.perf_counter() - t0
        bases = int(sum(hb["n_bases"] for hb in host_batches))
        reads = int(sum(len(hb["reads"]) for hb in host_batches))
        bases1 = int(sum(hb["n_bases"] for hb in host_batches[:n1]))
        common = ["-b"] + wl["cli"] + ["-K", str(args.batch), "-B", "200M"]

        def run(cmd, out_path, env=None):
            time.sleep(1.0)   # (the kernel -- or the CLI's exit helper -- is still clearing the last GPU process away for 0.1 - 0.3 s after it is reaped; a run started into that pays for it in its own start-up)
            t = time.perf_counter()
            r = subprocess.run(cmd + ["-o", out_path, fa] + [cmd_bam[0]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            wall = time.perf_counter() - t
            if r.returncode != 0:
                raise SystemExit("end-to-end run failed: %s\n%s" % (" ".join(cmd), r.stderr.decode(errors="replace")[-2000:]))
            return wall, r.stderr.decode(errors="replace")

        def digest(path):
            h = hashlib.md5()
            with open(path, "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            return h.hexdigest(), os.path.getsize(path)
        cmd_bam = [bam]
        out_gpu, out_cpu, out_cpu1 = os.path.join(tmp, "gpu.bed"), os.path.join(tmp, "cpu.bed"), os.path.join(tmp, "cpu1.bed")
        gpu_cmd = [cli, "freq"] + os.environ.get("MM_E2E_CLI_FLAGS", "").split() + common + ["-t", str(threads)]
        # the first run pages the file in and brings the HIP runtime up cold; then THREE timed runs, the median is the figure -- for the GPU and for the CPU
        # leg alike (round 6).  The wall is the caller's: spawn to reaped child, the kernel's clearing away of the GPU process included (the exit helper of
        # round 5 that left that behind the wait is opt-in now, MM_ASYNC_EXIT=1: one run with it is reported beside, wall_s_async_exit).
        w_first, err_gpu = run(gpu_cmd, out_gpu)
        gruns = sorted((run(gpu_cmd, out_gpu) for _ in range(3)), key=lambda x: x[0])
        w_gpu, err_gpu = gruns[1]
        w_async, _ = run(gpu_cmd, out_gpu, env=dict(os.environ, MM_ASYNC_EXIT="1"))
        cpu_cmd = [cpu_cli] + common + ["-t", str(threads)]
        cruns = sorted((run(cpu_cmd, out_cpu) for _ in range(3)), key=lambda x: x[0])
        w_cpu, err_cpu = cruns[1]
        cmd_bam = [bam1]
        w_cpu1, err_cpu1 = run([cpu_cli] + common + ["-t", "1"], out_cpu1)
        (md_g, sz_g), (md_c, sz_c) = digest(out_gpu), digest(out_cpu)
        e2e = {"value": bases / w_gpu / 1e6, "unit": "Mbases/s", "wall_s": w_gpu, "wall_s_runs": [g[0] for g in gruns], "wall_s_first_run": w_first, "wall_s_async_exit": w_async,
               "exit": "wall_s = median of three runs, spawn to reaped child, the process's teardown (GPU queues' save areas, pinned buffers) inside the caller's wait; "
                       "wall_s_async_exit = one run with MM_ASYNC_EXIT=1 (csrc/host/exitpath.c: a helper takes the address space apart behind the process)", "bases": bases,
               "reads": reads, "threads": threads, "cmd": "minimod freq " + " ".join(common + ["-t", str(threads)]) + " ref.fa reads.bam",
               "what": "whole child process: start, HIP initialisation, FASTA load + context kernel, BGZF/BAM decode of a %d MB file with "
                       "filter fodder, batches through host memory, finalize, %d MB of bedmethyl written" % (os.path.getsize(bam) >> 20, sz_g >> 20),
               "stages_s": _stage_timers(err_gpu), "bam_bytes": os.path.getsize(bam), "input_build_s": t_write,
               "parity_vs_cpu": {"byte_identical": md_g == md_c and sz_g == sz_c, "bytes": sz_g, "md5": md_g}}
        cpu = {"kind": "port", "unit": "Mbases/s", "cores": cores,
               "t_all": {"value": bases / w_cpu / 1e6, "wall_s": w_cpu, "wall_s_runs": [c[0] for c in cruns], "threads": threads, "bases": bases, "stages_s": _stage_timers(err_cpu),
                         "cmd": "oracle/_build/freq_cpu " + " ".join(common + ["-t", str(threads)]) + " ref.fa reads.bam"},
               "t_1": {"value": bases1 / w_cpu1 / 1e6, "wall_s": w_cpu1, "threads": 1, "bases": bases1, "stages_s": _stage_timers(err_cpu1),
                       "sample": "first %d of %d -K %d batches as their own BAM" % (n1, len(host_batches), args.batch)},
               "what": "oracle/freq_cpu_main.c: the oracle (bit-exact CPU restatement) behind the same BGZF/BAM reader, -K/-B batching and "
                       "row formatter as the product CLI, load(N+1) overlapping process(N) like the reference's pipeline"}
        return e2e, cpu
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process never touches
    the GPU), rank r on GPU r, rendezvous on 127.0.0.1.  Rank 0's JSON line is this process's output."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, p.wait())
    sys.stdout.write(out.decode(errors="replace"))
    sys.stdout.flush()
    if rc:
        raise SystemExit(rc)


def run_e2e_big(args):
    """`--e2e-gbases G`: a job long enough that start-up does not matter.  G Gbases of the workload's reads at 30x over G/30
    Gb of reference, written once as a BGZF BAM (filter fodder included) + FASTA; `minimod freq` (GPU) twice and the CPU port
    once on the same files, every run a child process; the bedmethyl outputs compared byte for byte.  Nothing here touches
    the GPU in this process."""
    from minimod_amd import synth
    from oracle import oracle as O
    wl = WORKLOADS[args.config]
    cores = usable_cores()
    threads = args.e2e_threads if args.e2e_threads > 0 else min(cores, 128)
    bases_target = args.e2e_gbases * 1e9
    region = max(1 << 20, int(bases_target / 30) // (1 << 20) * (1 << 20))
    n_reads = int(bases_target / 15070)
    plan = single_contig_plan(0, 1, region, HALO)
    t0 = time.perf_counter()
    refs = plan_references(plan, args.seed)
    iv = plan["intervals"][0]
    jobs = [(f, min(args.batch, n_reads - f)) for f in range(0, n_reads, args.batch)]

    def gen(job):
        first, n = job
        return synth.batch(refs[0], first, n, seed=args.seed, contig_len=plan["contigs"][0][1], n_reads_total=n_reads, tid=0,
                           region_begin=iv["read_begin"], region_len=iv["read_len"], with_order=False, **wl["gen"])
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > int(2.5 * bases_target) else None
    tmp = tempfile.mkdtemp(prefix="mm_e2e_big_", dir=base)
    try:
        bam, fa = os.path.join(tmp, "reads.bam"), os.path.join(tmp, "ref.fa")
        bases, bai_parts = 0, []
        # generated and written in rounds of 64 batches (memory stays bounded), the pieces concatenated
        with open(bam, "wb") as out:
            for r0 in range(0, len(jobs), 64):
                with ThreadPoolExecutor(max_workers=min(32, cores)) as ex:
                    bs = list(ex.map(gen, jobs[r0:r0 + 64]))
                bases += int(sum(b["n_bases"] for b in bs))
                piece = os.path.join(tmp, "piece.bam")
                pr = synth.write_bam_rounds(piece, plan["contigs"], bs, first_round=r0 == 0, last_round=r0 + 64 >= len(jobs), first_read=r0 * args.batch, threads=min(32, cores),
                                            index=bool(args.e2e_devices))
                bai_parts += [(x, out.tell() + at) for x, at in pr]
                with open(piece, "rb") as f:
                    shutil.copyfileobj(f, out, 1 << 24)
                os.remove(piece)
        if args.e2e_devices:   # the workers of a --devices run open the file where the index says their share begins
            synth.merge_bai(bam + ".bai", bai_parts)
        synth.write_fasta(fa, plan["contigs"][0][0], refs[0])
        t_build = time.perf_counter() - t0
        common = ["-b"] + wl["cli"] + ["-K", str(args.batch), "-B", "200M", "-t", str(threads)]
        cli = os.path.join(ROOT, "minimod_amd", "bin", "minimod")
        cpu_cli = O.build_cpu_cli()

        def run(cmd, out_path, env=None):
            time.sleep(float(os.environ.get("MM_E2E_PAUSE", "1.5")))   # (the kernel goes on clearing a HIP process away after it is reaped: a run started right behind another waits for that in its own start-up, 0.2 - 0.3 s)
            t, e0 = time.perf_counter(), time.time()
            r = subprocess.run(cmd + ["-o", out_path, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            wall, e1 = time.perf_counter() - t, time.time()
            if r.returncode != 0:
                raise SystemExit("end-to-end run failed: %s\n%s" % (" ".join(cmd), r.stderr.decode(errors="replace")[-2000:]))
            err = r.stderr.decode(errors="replace")
            if os.environ.get("MM_TIMELINE"):   # what lies outside the CLI's own clock: spawn -> its first mark, its last mark -> reaped
                marks = [float(x) for x in re.findall(r"\(epoch ([0-9.]+),", err)]
                if marks:
                    err += "[outside] %.3f s from the spawn to the first mark, %.3f s from the last mark to the reaped child\n" % (marks[0] - e0, e1 - marks[-1])
            return wall, err

        def md5(path):
            h = hashlib.md5()
            with open(path, "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            return h.hexdigest()
        og, oc = os.path.join(tmp, "gpu.bed"), os.path.join(tmp, "cpu.bed")
        # rows can tie on (contig, start) with several codes / --insertions / --haplotypes: the CPU port prints them in the fixed
        # order, so the compared GPU run does too; the default run -- the reference's order, replayed on the host -- is timed beside it
        tied = len(wl["mods"]) > 1 or bool(wl["eng"])
        gpu_flags = os.environ.get("MM_E2E_CLI_FLAGS", "").split()   # e.g. --gpu-inflate (the GPU CLI's runs only)
        # one cold run, then three timed ones: the median, teardown inside the caller's wait (round 6; MM_ASYNC_EXIT=1 -- the exit helper -- once beside it)
        gcmd = [cli, "freq"] + gpu_flags + (["--canonical-order"] if tied else []) + common
        first = run(gcmd, og)
        runs = sorted((run(gcmd, og) for _ in range(3)), key=lambda x: x[0])
        wall, err = runs[1]
        w_async, _ = run(gcmd, og, env=dict(os.environ, MM_ASYNC_EXIT="1"))
        if os.environ.get("MM_E2E_STDERR"):   # the CLI's own log of the timed run (its lines carry the time since start)
            with open(os.environ["MM_E2E_STDERR"], "w") as f:
                f.write(err)
        st = _stage_timers(err)
        startup = st.get("reference", 0.0) + st.get("contexts", 0.0) + st.get("gpu_runtime", 0.0)
        m = re.search(r"GPU launches: (\d+) for (\d+) batches \((\d+) with k_stream_reads\)", err)
        cruns = sorted((run([cpu_cli] + common, oc) for _ in range(3)), key=lambda x: x[0])
        w_cpu, err_cpu = cruns[1]
        st_cpu = _stage_timers(err_cpu)
        res = {"metric": "minimod freq end to end, steady state", "unit": "Mbases/s", "bases": bases, "reads": n_reads, "reference_bases": region,
               "bam_bytes": os.path.getsize(bam), "threads": threads, "cores": cores, "input_build_s": t_build,
               "gpu_cli": {"value": bases / wall / 1e6, "value_without_startup": bases / max(wall - startup, 1e-9) / 1e6, "wall_s": wall,
                           "wall_s_runs": [r_[0] for r_ in runs], "wall_s_first_run": first[0], "wall_s_async_exit": w_async, "startup_s": startup, "stages_s": st,
                           "exit": "wall_s: median of three, the teardown inside the caller's wait; wall_s_async_exit: MM_ASYNC_EXIT=1, a helper takes the address space apart behind the process (csrc/host/exitpath.c)",
                           "launches": {"launches": int(m.group(1)), "batches": int(m.group(2)), "with_k_stream_reads": int(m.group(3))} if m else None,
                           "cmd": "minimod freq " + " ".join(gpu_flags + common) + " ref.fa reads.bam",
                           "what": "whole child process (start, HIP initialisation, FASTA load + context kernels, BGZF/BAM decode, batches through "
                                   "mm_freq_submit, finalize, bedmethyl written); value_without_startup leaves out the reference load and context "
                                   "kernels (the part that does not grow with the reads)"},
               "cpu_port": {"kind": "port", "value": bases / w_cpu / 1e6, "value_without_startup": bases / max(w_cpu - st_cpu.get("reference", 0.0) - st_cpu.get("contexts", 0.0), 1e-9) / 1e6,
                            "wall_s": w_cpu, "wall_s_runs": [c_[0] for c_ in cruns], "stages_s": st_cpu, "cmd": "oracle/_build/freq_cpu " + " ".join(common) + " ref.fa reads.bam"},
               "parity_vs_cpu": {"byte_identical": md5(og) == md5(oc) and os.path.getsize(og) == os.path.getsize(oc), "bytes": os.path.getsize(og)}}
        ml = re.search(r"\[loader\] ([^\n]*)", err)
        if ml:
            res["gpu_cli"]["loader"] = ml.group(1)
        mz = re.search(r"\[gpu-inflate\] ([^\n]*)", err)   # (the CLI's default for a BAM of 3 GiB or more per GPU; --no-gpu-inflate: host threads alone)
        res["gpu_cli"]["gpu_inflate"] = mz.group(1) if mz else None
        mi = re.search(r"\[gpu-ingest\] ([^\n]*)", err)   # (the default for a tie-free run on a BAM of 1 GiB or more per GPU: the decoded stream stays on the device)
        res["gpu_cli"]["gpu_ingest"] = mi.group(1) if mi else None
        # MM_E2E_VARIANTS="host:--no-gpu-ingest --no-gpu-inflate;inflate:--no-gpu-ingest --gpu-inflate": the same files through the GPU CLI
        # with other flags (two runs each, the faster one kept; MM_E2E_STDERR gets ".<name>" appended for the CLI's log)
        for spec in [x for x in os.environ.get("MM_E2E_VARIANTS", "").split(";") if x.strip()]:
            name, _, fl = spec.partition(":")
            ov = os.path.join(tmp, "gpu_var.bed")
            vruns = [run([cli, "freq"] + fl.split() + (["--canonical-order"] if tied else []) + common, ov) for _ in range(2)]
            vw, verr = min(vruns, key=lambda x: x[0])
            if os.environ.get("MM_E2E_STDERR"):
                with open(os.environ["MM_E2E_STDERR"] + "." + name.strip(), "w") as f:
                    f.write(verr)
            res.setdefault("variants", {})[name.strip()] = {"flags": fl.strip(), "wall_s": vw, "wall_s_first_run": vruns[0][0], "value": bases / vw / 1e6, "stages_s": _stage_timers(verr),
                                                            "byte_identical_to_cpu": md5(ov) == md5(oc)}
        # MM_E2E_ENV_VARIANTS="eager:MM_INGEST_EAGER_SLOTS=1": the same files and flags with other environment variables, three alternating
        # pairs of runs (default, variant) on this box -- an A/B that box-to-box differences do not drown
        for spec in [x for x in os.environ.get("MM_E2E_ENV_VARIANTS", "").split(";") if x.strip()]:
            name, _, ev = spec.partition(":")
            venv = dict(os.environ, **dict(kv.split("=", 1) for kv in ev.split()))
            ov = os.path.join(tmp, "gpu_env.bed")
            pairs = []
            for _ in range(3):
                wa, ea = run([cli, "freq"] + gpu_flags + (["--canonical-order"] if tied else []) + common, ov)
                wb, eb = run([cli, "freq"] + gpu_flags + (["--canonical-order"] if tied else []) + common, ov, env=venv)
                def outside(e):   # (MM_TIMELINE=1: spawn -> first mark, last mark -> reaped, and the CLI's own clock)
                    m1, m2 = re.search(r"\[outside\] ([0-9.]+) s from the spawn to the first mark, ([0-9.]+) s", e), re.search(r"Real time: ([0-9.]+) sec", e)
                    return {"before_main_s": float(m1.group(1)), "after_last_word_s": float(m1.group(2)), "real_time_s": float(m2.group(1)) if m2 else None} if m1 else None
                pairs.append({"default": {"wall_s": wa, "stages_s": _stage_timers(ea), "outside": outside(ea)}, name.strip(): {"wall_s": wb, "stages_s": _stage_timers(eb), "outside": outside(eb)}})
            res.setdefault("env_variants", {})[name.strip()] = {"env": ev.strip(), "pairs": pairs, "byte_identical_to_cpu": md5(ov) == md5(oc)}
        if args.e2e_devices:
            # one BAM, N worker processes (csrc/host/freq_main.c run_devices: shares cut at 64 kb-aligned positions, every worker reads its share
            # through the index, counts on its GPU, hands its halo slab to the right-hand neighbour, formats its own section)
            od = os.path.join(tmp, "gpu_devices.bed")
            druns = [run([cli, "freq", "--devices", args.e2e_devices] + gpu_flags + (["--canonical-order"] if tied else []) + common, od) for _ in range(2)]
            dw, derr = min(druns, key=lambda x: x[0])
            if os.environ.get("MM_E2E_STDERR"):
                with open(os.environ["MM_E2E_STDERR"] + ".devices", "w") as f:
                    f.write(derr)
            workers = [{"worker": int(m.group(1)), "device": int(m.group(2)), "reads": int(m.group(3)), "mbases": float(m.group(4)), "load_s": float(m.group(5)),
                        "waiting_for_gpu_s": float(m.group(6)), "finalize_s": float(m.group(7))}
                       for m in re.finditer(r"worker (\d+) \(device (\d+)\): (\d+) entries, ([0-9.]+) Mbases, loading ([0-9.]+) sec, waiting for the GPU ([0-9.]+) sec, finalize ([0-9.]+) sec", derr)]
            sd = _stage_timers(derr)
            res["devices_run"] = {"devices": args.e2e_devices, "n_workers": len(args.e2e_devices.split(",")), "wall_s": dw, "wall_s_first_run": druns[0][0], "value": bases / dw / 1e6,
                                  "workers": workers, "parent_merge_s": sd.get("merge"), "parent_output_s": sd.get("output"), "stages_s": sd,
                                  "slabs_through_hip_ipc": len(re.findall(r"through a HIP IPC handle", derr)),
                                  "byte_identical_to_single_run": md5(od) == md5(og) and os.path.getsize(od) == os.path.getsize(og),
                                  "cmd": "minimod freq --devices %s " % args.e2e_devices + " ".join(gpu_flags + common) + " ref.fa reads.bam",
                                  "what": "the same files through `minimod freq --devices`: the parent never touches a GPU, forks one worker per listed device; workers listed on "
                                          "the same device share it (a one-GPU box shows the path and its costs, not scaling)"}
        # MM_E2E_SWEEP="16,32,64": the same job at other -t (diagnostic: where the host side stops scaling)
        for t_alt in [int(x) for x in os.environ.get("MM_E2E_SWEEP", "").split(",") if x.strip()]:
            alt = ["-b"] + wl["cli"] + ["-K", str(args.batch), "-B", "200M", "-t", str(t_alt)]
            w_alt, err_alt = run([cli, "freq"] + (["--canonical-order"] if tied else []) + alt, os.path.join(tmp, "gpu_alt.bed"))
            ma = re.search(r"\[loader\] ([^\n]*)", err_alt)
            res.setdefault("threads_sweep", []).append({"threads": t_alt, "wall_s": w_alt, "stages_s": _stage_timers(err_alt), "loader": ma.group(1) if ma else None})
        if tied:
            w_rp, err_rp = min((run([cli, "freq"] + gpu_flags + common, os.path.join(tmp, "gpu_replay.bed")) for _ in range(2)), key=lambda x: x[0])
            if os.environ.get("MM_E2E_STDERR"):
                with open(os.environ["MM_E2E_STDERR"] + ".replay", "w") as f:
                    f.write(err_rp)
            mr = re.search(r"Row order replay[^:]*: ([0-9.]+) sec", err_rp)
            mrf = re.search(r"Row order replay[^:]*: [0-9.]+ sec \(([0-9.]+) of them waiting", err_rp)
            mp = re.search(r"Peak RAM: ([0-9.]+) GB", err_rp)
            mp0 = re.search(r"Peak RAM: ([0-9.]+) GB", err)
            res["reference_order_replay"] = {"wall_s": w_rp, "value": bases / w_rp / 1e6, "replay_s": float(mr.group(1)) if mr else None, "replay_waiting_for_gpu_s": float(mrf.group(1)) if mrf else None, "stages_s": _stage_timers(err_rp),
                                             "peak_ram_gb": float(mp.group(1)) if mp else None, "peak_ram_gb_canonical": float(mp0.group(1)) if mp0 else None,
                                             "same_rows_as_canonical": sorted(open(os.path.join(tmp, "gpu_replay.bed"), "rb").read().splitlines()) == sorted(open(og, "rb").read().splitlines())
                                             if os.path.getsize(og) < (1 << 30) else None,
                                             "replay_on": "device" if "on the device" in err_rp else "host",
                                             "gpu_ingest": bool(re.search(r"\[gpu-ingest\] ", err_rp)),
                                             "what": "the default run of a tied configuration: a second handle in view mode leaves every call in GPU memory, the device replays the "
                                                     "reference's per-read tables, core hash table and sort (csrc/tie_kernels.hip.h); --canonical-order skips it"}
            # round 4's path, the serial restatement on the host (csrc/host/tieorder.c): the checker -- same bytes -- and the time it took
            w_hr, err_hr = min((run([cli, "freq", "--host-replay"] + gpu_flags + common, os.path.join(tmp, "gpu_host_replay.bed")) for _ in range(2)), key=lambda x: x[0])
            mh = re.search(r"Row order replay[^:]*: ([0-9.]+) sec", err_hr)
            res["reference_order_replay"]["host_replay"] = {"wall_s": w_hr, "replay_s": float(mh.group(1)) if mh else None,
                                                            "byte_identical_to_device_replay": md5(os.path.join(tmp, "gpu_host_replay.bed")) == md5(os.path.join(tmp, "gpu_replay.bed"))}
        print(json.dumps(res))
        sys.stdout.flush()
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = parse_args()
    if args.e2e_gbases > 0:
        return run_e2e_big(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MM_BENCH_LAUNCH_ONLY"):   # the CPU test of the self-launch: what a rank was started with, nothing else
        if rank == 0:
            print(json.dumps({"launch_only": True, "n_gpus": world, "rank": rank, "local_rank": local_rank, "master": os.environ.get("MASTER_ADDR")}))
        return None
    wl = WORKLOADS[args.config]
    if "reads" in wl and args.reads == 100000:   # (C5, C4: the workload's own number of reads per GPU)
        args.reads = wl["reads"]
    import torch                      # importing torch does not initialise the GPU
    import torch.distributed as dist

    import minimod_amd
    from minimod_amd import engine, synth

    # N = 1: one contig (C2 as BASELINE.json states it).  N > 1: the 24-contig genome, one contiguous share per rank.
    region = wl.get("region", INTERVAL) if args.region_mb <= 0 else int(args.region_mb * (1 << 20)) // (1 << 20) * (1 << 20)
    if (world == 1 and not wl.get("genome")) or args.single_contig:
        plan = single_contig_plan(rank, world, region, HALO)
    else:
        plan = genome_plan(rank, world, genome_layout(world, region), HALO)
    if wl.get("genome") and world == 1:
        args.no_e2e = True   # (the end-to-end legs write the whole read set as a BAM: --e2e-gbases is the tool for that size)
    t0 = time.time()
    refs = plan_references(plan, args.seed)
    names = [n for n, _ in plan["contigs"]]
    share = sum(iv["read_len"] for iv in plan["intervals"])
    jobs, left = [], args.reads      # (interval ordinal, first read, reads): pieces of at most -K reads, generated in parallel
    for k, iv in enumerate(plan["intervals"]):
        n_iv = left if k == len(plan["intervals"]) - 1 else min(left, int(round(args.reads * iv["read_len"] / share)))
        left -= n_iv
        jobs += [(k, f, min(args.batch, n_iv - f), n_iv) for f in range(0, n_iv, args.batch)]

    def gen(job):
        k, first, n, n_iv = job
        iv = plan["intervals"][k]
        return synth.batch(refs[iv["tid"]], first, n, seed=args.seed + 7919 * rank + 101 * k, contig_len=plan["contigs"][iv["tid"]][1],
                           n_reads_total=n_iv, tid=iv["tid"], region_begin=iv["read_begin"], region_len=iv["read_len"], max_len=args.max_len,
                           with_order=False, **wl["gen"])

    with ThreadPoolExecutor(max_workers=min(16, usable_cores())) as ex:
        pieces = list(ex.map(gen, jobs))
    whole = synth.concat(pieces)
    # (N = 1: the pieces are the -K batches themselves, each with pools of its own -- what a host caller hands to mm_freq_submit)
    compact_batches = pieces if (world == 1 and all(len(pc["reads"]) == min(args.batch, len(whole["reads"]) - i * args.batch) for i, pc in enumerate(pieces))) else None
    del pieces
    n_reads = len(whole["reads"])
    host_batches = synth.split(whole, [min(args.batch, n_reads - i) for i in range(0, n_reads, args.batch)])   # the -K windows
    n_batches = len(host_batches)
    t_gen = time.time() - t0

    # ---- the other BASELINE workloads' roofline fractions: each a child process of its own (this same script, no CPU legs), started
    # before this process has touched the GPU
    config_fracs = None
    if world == 1 and args.mode == "freq" and args.config == "C2" and not args.no_extra and not args.no_config_fracs and torch.cuda.device_count() > 0:
        config_fracs = {}
        for name, extra, st in (("C3", ["--config", "C3"], 20), ("C5", ["--config", "C5"], 17), ("view", ["--mode", "view"], 20)):
            t_c = time.time()
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra + ["--steps", str(st), "--warmup", "5", "--reps", "3", "--no-e2e", "--no-cpu-baseline", "--no-extra"],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
                d = json.loads(r.stdout.decode().strip().splitlines()[-1])
                rf = d["roofline"]
                config_fracs[name] = {"frac": rf["frac"], "achieved": rf["achieved"], "traffic": rf.get("traffic"), "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                                      "kernel_ms_per_batch": rf.get("kernel_ms_per_batch"), "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"), "steps": st, "workload": d["config"]["workload"], "wall_s": time.time() - t_c}
            except Exception as e:   # (a leg that fails leaves its reason, never a number)
                config_fracs[name] = {"frac": None, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    # ---- the end-to-end leg runs in child processes, before this process has touched the GPU
    e2e = cpu_e2e = None
    if world == 1 and args.mode == "freq" and not args.no_e2e:
        e2e, cpu_e2e = run_end_to_end(args, wl, host_batches, plan["contigs"][0], refs[0])

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

    contig = [(n, l, refs[t]) for t, (n, l) in enumerate(plan["contigs"])]
    if args.mode == "view":
        eng = minimod_amd.FreqEngine(wl["mods"], contig, device=local_rank, view=True, coalesce=args.coalesce, **wl["eng"])
    else:
        eng = minimod_amd.FreqEngine(wl["mods"], contig, device=local_rank,
                                     intervals=[(iv["tid"], iv["begin"], iv["end"], iv["halo"]) for iv in plan["intervals"]],
                                     side_capacity=(96 << 20) if wl["eng"].get("insertions") else 0, split_bases=args.split_bases, force_fused=args.force_fused, coalesce=args.coalesce, stream_mode=1 if args.no_stream else 0, stream_slices=args.stream_slices,
                                     **wl["eng"])
    # ---- make the reads resident in HBM (torch owns the memory: plumbing only): ONE read set -- the pools of all batches
    # end to end, as a decoder writing into device memory would leave them -- and a step's batch is a window of -K reads of
    # it (the same pool pointers, `reads` advanced).  Consecutive windows are what mm_freq_opts_t.coalesce may gather.
    keep, base = [], {}
    for k in ("reads", "cigar", "seq", "mm", "ml"):
        t = torch.from_numpy(whole[k].view(np.uint8).reshape(-1)).to(dev)
        keep.append(t)
        base[k] = t.data_ptr()
    dev_batches, first = [], 0
    for hb in host_batches:
        n = len(hb["reads"])
        d = dict(reads=base["reads"] + 64 * first, cigar=base["cigar"], seq=base["seq"], mm=base["mm"], ml=base["ml"],
                 n_reads=n, n_cigar_words=len(whole["cigar"]), n_seq_bytes=len(whole["seq"]), n_mm_bytes=len(whole["mm"]),
                 n_ml_bytes=len(whole["ml"]), max_n_cigar=hb["max_n_cigar"], max_l_qseq=hb["max_l_qseq"])
        if args.host_plan:
            items = engine.plan_batch(hb["reads"])
            t = torch.from_numpy(items.view(np.uint8).reshape(-1)).to(dev)
            keep.append(t)
            d["order"] = t.data_ptr()
            d["n_order"] = len(items)
        dev_batches.append(d)
        first += n
    pool_sizes = {k: len(whole[k]) for k in ("cigar", "seq", "mm", "ml")}
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(device=dev)   # one explicit HIP stream carries every K1 launch
    stream = tstream.cuda_stream
    batch_bases = [hb["n_bases"] for hb in host_batches]

    if args.mode == "view":
        return bench_view(args, eng, host_batches, dev_batches, batch_bases, stream, rank, world, dist, dev, plan, refs, t_gen)

    # ---- untimed tally pass: lookups/updates per batch for the algorithmic-bytes figure
    eng.stats_enable(True)
    alg_bytes, side_per_pass = [], 0
    routing, phases = np.zeros(3, dtype=np.int64), np.zeros(12, dtype=np.int64)
    for hb, db in zip(host_batches, dev_batches):
        eng.wait(eng.submit_device(db, stream))
        st = eng.stats_get()
        routing += np.array([st["stream_done"], st["stream_to_tiles"], st["stream_to_fused"]], dtype=np.int64)
        phases += np.array(st["phase_cycles"], dtype=np.int64)
        side_per_pass += st["side_updates"]
        alg_bytes.append(algorithmic_bytes(hb["reads"], st["lookups"], st["dense_updates"] + st["side_updates"]))
    # ... and where the reads go in launches like the timed ones (the windows gathered as the timed steps gather them: which
    # reads k_stream_reads takes depends on the size of the launch)
    routing[:] = 0
    last = None
    for db in dev_batches[:max(1, min(n_batches, args.steps))]:
        last = eng.submit_device(db, stream)
    eng.wait(last)
    st = eng.stats_get()
    routing += np.array([st["stream_done"], st["stream_to_tiles"], st["stream_to_fused"]], dtype=np.int64)
    routed_reads = int(sum(len(hb["reads"]) for hb in host_batches[:max(1, min(n_batches, args.steps))]))
    eng.stats_enable(False)
    eng.reset()
    # the side list (calls inside insertions: --insertions only) grows with every step and is only emptied by reset()
    side_cap = 96 << 20
    if side_per_pass * ((args.steps + n_batches - 1) // n_batches + 1) > side_cap:
        raise SystemExit("--config %s: %d side-list records per pass over the batches: too many steps for a list of %d"
                         % (args.config, side_per_pass, side_cap))

    from minimod_amd import engine as _E
    dev_structs = [_E.batch_struct(db, device=True) for db in dev_batches]   # (the C structs of the resident windows, made once: a C caller has them anyway)

    def run_steps(n, first_step=0, use_stream=stream):
        """n steps = n -K windows submitted in order; returns bases, device time per LAUNCH (a launch carries one window, or
        up to --coalesce consecutive ones: submits of one group return the same ticket), algorithmic bytes."""
        groups, bases, kms, abytes = [], 0, [], 0

        def retire(g):
            eng.wait(g[0])
            kms.append(eng.kernel_ms(g[0]))
        for s in range(n):
            bi = (first_step + s) % n_batches
            t = eng.submit_device(dev_structs[bi], use_stream)
            if groups and groups[-1][0] == t:
                groups[-1][1] += 1
            else:
                groups.append([t, 1])
            bases += batch_bases[bi]
            abytes += alg_bytes[bi]
            if len(groups) > 3:   # the library has 4 slots; keep the host a few launches ahead of the device
                retire(groups.pop(0))
        for g in groups:
            retire(g)
        return bases, kms, abytes

    # ---- warm-up, then the timed region: barrier + sync on both sides, max over ranks
    run_steps(args.warmup)
    eng.reset()
    slab_len = max([x[2] for x in (plan["send"], plan["recv"]) if x] + [0])
    if world > 1:   # every slab of the job has the same size (the halo); ranks without a cut on one side skip that side
        tl = torch.tensor([slab_len], dtype=torch.int64, device="cpu" if host_staged else dev)
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        slab_len = int(tl.item())
    slab_words = eng.slab_words(slab_len)

    def make_buf():
        t = torch.zeros(slab_words, dtype=torch.int64, device="cpu" if host_staged else dev)
        torch.cuda.synchronize()   # (the fill runs on torch's current stream, the slab kernels on `stream`: not ordered otherwise)
        return t

    def export_fn(buf):
        tid, pos, ln = plan["send"]
        dbuf = torch.zeros(slab_words, dtype=torch.int64, device=dev) if host_staged else buf
        torch.cuda.synchronize()
        eng.slab_export(tid, pos, ln, dbuf.data_ptr(), stream)
        eng.slab_clear(tid, pos, ln, stream)
        tstream.synchronize()          # the slab must be complete before RCCL (another stream) reads it
        if host_staged:
            buf.copy_(dbuf.cpu())

    def add_fn(buf):
        tid, pos, ln = plan["recv"]
        dbuf = buf.to(dev) if host_staged else buf
        torch.cuda.synchronize()       # the received slab is complete before our stream adds it
        eng.slab_add(tid, pos, ln, dbuf.data_ptr(), stream)
        tstream.synchronize()

    def exchange():
        exchange_slabs(rank, world, plan["send"] is not None, plan["recv"] is not None, export_fn, add_fn, make_buf, dist)

    if world > 1:
        # warm the point-to-point path too (RCCL sets up its send/recv channels on first use: that must not land in
        # the timed region), then start from clean counters again
        exchange()
        eng.reset()
        dist.barrier()
    # The timed region, --reps times over: each repetition is EXACTLY --steps steps between barrier + synchronize on both
    # sides (max over ranks), from clean counters; `value` is the median repetition, `spread` says how far they lie apart.
    reps = []
    for rep in range(max(1, args.reps)):
        if rep:
            eng.reset()
            if world > 1:
                dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bases, kms, abytes = run_steps(args.steps, first_step=args.warmup, use_stream=stream if args.streams <= 1 else None)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        reps.append({"elapsed": time.perf_counter() - t0, "kms": kms, "abytes": abytes})
    # The job's one exchange: each rank's halo slab goes to its right neighbour after the LAST batch (once per job, not per
    # step: a 30x genome is thousands of steps).  It runs here, right behind the last repetition's K timed steps, and is timed on its own.
    t2 = time.perf_counter()
    exchange()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    reduce_s = time.perf_counter() - t2
    el = [r["elapsed"] for r in reps]
    if world > 1:
        rdev = "cpu" if host_staged else dev
        tt = torch.tensor(el + [reduce_s], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el, reduce_s = [float(x) for x in tt[:-1].tolist()], float(tt[-1].item())
        tb = torch.tensor([bases], dtype=torch.int64, device=rdev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases = int(tb.item())
        # every rank's own figures (its bases, the seconds of its median repetition), for the line's per_rank
        mine = torch.tensor([float(bases), float(sorted(r["elapsed"] for r in reps)[len(reps) // 2])], dtype=torch.float64, device=rdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": k, "bases": int(v[0].item()), "seconds": float(v[1].item()), "Mbases_per_s": float(v[0].item()) / max(float(v[1].item()), 1e-12) / 1e6} for k, v in enumerate(allr)]
    else:
        total_bases = bases
        per_rank = None
    mid = int(np.argsort(el)[len(el) // 2])      # the median repetition (the upper one of an even count)
    elapsed, kms, abytes = el[mid], reps[mid]["kms"], reps[mid]["abytes"]

    # extra legs, outside the contract's timed region (N = 1 only): the same steps one launch per step, and through the
    # product's entry point (host batches, PCIe included)
    legs = None
    if world == 1 and not args.no_extra and not args.no_host_path and not args.dump_timed and compact_batches is not None:
        legs = extra_legs(args, wl, contig, plan, compact_batches, dev_batches, batch_bases, alg_bytes, stream, local_rank)

    result = None
    if rank == 0:
        mean_ms = float(np.mean(kms))
        achieved = abytes / (float(np.sum(kms)) * 1e-3) / 1e9
        # HBM traffic per batch comes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) of this same command, which cannot
        # run inside it: the figure is read from the committed profile and labelled as such
        traffic, traffic_src = committed_traffic(args.config.lower(), args.steps / len(kms))   # per launch, like `achieved`
        reads_all = np.concatenate([hb["reads"]["l_qseq"] for hb in host_batches])
        result = {
            "metric": "minimod freq Mbases/sec", "value": total_bases / elapsed / 1e6, "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": wl["what"] % dict(reads=args.reads, mb=region / 1e6, batch=args.batch),
                       "reads_per_gpu": args.reads, "batch_reads": args.batch, "mean_read_len": int(np.mean(reads_all)),
                       "sharding": "single GPU, one contig" if (world == 1 and not wl.get("genome")) else
                                   ("one long contig, one interval per GPU + halo slab to the right neighbour" if args.single_contig else
                                    "24 contigs with hg38's length ratios, %.1f Mb in all, cut into one contiguous share per GPU at 64 kb-aligned "
                                    "positions; a halo slab goes to the right neighbour where a cut falls inside a contig (this rank: %d intervals, "
                                    "send %s, receive %s)" % (sum(l for _, l in plan["contigs"]) / 1e6, len(plan["intervals"]),
                                                               "yes" if plan["send"] else "no", "yes" if plan["recv"] else "no")),
                       "read_order": "caller's plan (mm_freq_plan_batch on the host, uploaded before the timed region)" if args.host_plan else
                                     "planned on the device inside every step (k_plan_items: long reads cut into parts, costliest first)",
                       "coalesce": "up to %d consecutive -K windows of the resident read set per launch (mm_freq_opts_t.coalesce)" % args.coalesce
                                   if args.coalesce > 1 and not args.host_plan else "off: one launch per step"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "freq hot path per batch = k_plan_items + k_stream_reads (reads that are one work item) + k_scan_reads + k_sum_tiles + "
                                   "k_call_tiles (the others) (+ k_freq_reads on the fallback list), HIP events around the launches on their stream",
                         "kernel_ms_mean": mean_ms, "algorithmic_bytes_per_launch": abytes / len(kms), "launches": len(kms),
                         "batches_per_launch": args.steps / len(kms), "kernel_ms_per_batch": float(np.sum(kms)) / args.steps,
                         "bytes_per_base": abytes / max(bases, 1), "side_list_updates_per_pass": side_per_pass},
            "gen_seconds": t_gen,
        }
        fr = [r["abytes"] / (float(np.sum(r["kms"])) * 1e-3) / 1e9 / HBM_PEAK_GBS for r in reps]
        result["spread"] = {"reps": len(reps), "what": "the timed region repeated: each repetition exactly %d steps between barrier + synchronize; value / ms_per_step / roofline "
                                                       "are the median repetition's" % args.steps,
                            "value": {"median": total_bases / elapsed / 1e6, "min": total_bases / max(el) / 1e6, "max": total_bases / min(el) / 1e6},
                            "roofline_frac": {"median": float(np.median(fr)), "min": float(min(fr)), "max": float(max(fr))},
                            "timed_region_ms": {"median": elapsed * 1e3, "min": min(el) * 1e3, "max": max(el) * 1e3}}
        result["config"]["routing"] = {"reads_done_by_k_stream_reads": int(routing[0]), "handed_to_the_tile_pipeline": int(routing[1]),
                                       "handed_to_the_fused_kernel": int(routing[2]), "reads": routed_reads,
                                       "note": "one pass over the first min(steps, batches) windows, gathered like the timed steps; reads too long "
                                               "to hide in the launch are the tile pipeline's from the start (all of them in a single-batch launch)"}
        timed_phases = eng.stats_get()["phase_cycles"]
        if phases[3:].any() or any(timed_phases[3:]):   # diagnostic builds (-DMM_STREAM_TIMING): the tally pass, then the timed steps
            result["phase_cycles"] = [int(x) for x in phases]
            result["phase_cycles_timed"] = [int(x) for x in timed_phases]
        if world > 1:
            result["final_reduce"] = {"ms": reduce_s * 1e3, "value_incl": total_bases / (elapsed + reduce_s) / 1e6, "unit": "Mbases/s",
                                      "ranks": world, "ranks_seen": dist.get_world_size(), "slab_bytes": slab_words * 8, "backend": args.backend,
                                      "per_rank": per_rank, "sends": int(plan["send"] is not None), "receives": int(plan["recv"] is not None),
                                      "note": "halo slabs (%d positions x planes x 8 B) to the right neighbour wherever a cut falls inside a "
                                              "contig, once per job after the last step; timed on its own, max over ranks; value_incl = "
                                              "throughput if these K steps were the whole job" % slab_len}
        if legs:
            result.update(legs)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, wl, host_batches, plan, refs)
        if config_fracs:
            # FIRST in the line (VERDICT rounds 4 and 5: a record that keeps a line's head or tail must keep these): every BASELINE workload's roofline
            # fraction and HBM traffic over algorithmic bytes, this run's; the detail rides further back
            short = {"C2": {"frac": round(result["roofline"]["frac"], 4), "traffic_x": (round(result["roofline"]["traffic"] / result["roofline"]["algorithmic_bytes_per_launch"], 2) if result["roofline"].get("traffic") else None)}}
            for k, v in config_fracs.items():
                short[k] = ({"frac": round(v["frac"], 4), "traffic_x": (round(v["traffic"] / v["algorithmic_bytes_per_launch"], 2) if v.get("traffic") and v.get("algorithmic_bytes_per_launch") else None)}
                            if v.get("frac") is not None else {"frac": None})
            result = dict([("config_fracs", short)] + list(result.items()))
            result["config_fracs_detail"] = config_fracs
            # (the driver's record keeps `roofline` whole and drops keys it does not know: the other workloads' fractions ride there too)
            result["roofline"]["other_workloads"] = {k: ({"frac": round(v["frac"], 4), "traffic_x": (round(v["traffic"] / v["algorithmic_bytes_per_launch"], 2)
                                                                                                        if v.get("traffic") and v.get("algorithmic_bytes_per_launch") else None)}
                                                         if v.get("frac") is not None else {"frac": None}) for k, v in config_fracs.items()}
        if e2e:
            result["end_to_end"] = e2e
            result["cpu_baseline_e2e"] = cpu_e2e
        if args.dump_timed:
            np.savez(args.dump_timed, rows=eng.finalize())
            result["dump_timed"] = {"file": args.dump_timed, "steps": args.steps, "warmup": args.warmup, "batches": n_batches}
        if args.dump:
            chk = minimod_amd.FreqEngine(wl["mods"], contig, device=local_rank, **wl["eng"])
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


def committed_traffic(name, batches_per_unit):
    """HBM bytes per `batches_per_unit` batches from profiles/traffic_<name>.json -- rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) of this
    same command, which cannot run inside it (tools/traffic.sh) -- or None when the file was measured on other kernel sources."""
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % name)
    if not os.path.exists(tpath):
        return None, None
    try:
        from minimod_amd.build import source_hash
        tj = json.load(open(tpath))
        if tj.get("source_hash") == source_hash():
            return (tj.get("hbm_bytes_per_batch") * batches_per_unit,
                    "profiles/%s: bytes per batch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command (not measured in this run; made from "
                    "these very kernel sources: source_hash %s), times this run's batches per launch" % (os.path.basename(tpath), tj.get("source_hash")))
        return None, ("profiles/%s was measured on other kernel sources (its source_hash %s, now %s): no traffic figure"
                      % (os.path.basename(tpath), tj.get("source_hash"), source_hash()))
    except Exception:
        return None, None


def extra_legs(args, wl, contig, plan, host_batches, dev_batches, batch_bases, alg_bytes, stream, local_rank):
    """Two more readings of the same K steps, each on its own handle: `resident_coalesce1` = the resident windows one launch per
    step (what a caller gets that waits for every -K batch); `host_path` = the batches handed over from HOST memory through
    mm_freq_submit -- the call the CLI and the INTEGRATION.md stub make -- which stages them in device memory and gathers them
    into launches like the timed region's (wall clock includes the PCIe copies; the kernel time is the launches' HIP events)."""
    import torch
    import minimod_amd
    n_batches = len(host_batches)
    common = dict(device=local_rank, intervals=[(iv["tid"], iv["begin"], iv["end"], iv["halo"]) for iv in plan["intervals"]],
                  side_capacity=(96 << 20) if wl["eng"].get("insertions") else 0, stream_mode=1 if args.no_stream else 0, **wl["eng"])
    steps = [(args.warmup + s) % n_batches for s in range(args.steps)]
    bases = int(sum(batch_bases[b] for b in steps))
    abytes = int(sum(alg_bytes[b] for b in steps))
    out = {}
    # ---- one launch per step
    e1 = minimod_amd.FreqEngine(wl["mods"], contig, coalesce=1, **common)
    for b in steps:   # (every step once untimed: a kernel a later window is the first to need is loaded on its first launch)
        e1.wait(e1.submit_device(dev_batches[b], stream))
    e1.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = []
    for b in steps:
        tk = e1.submit_device(dev_batches[b], stream)
        e1.wait(tk)
        kms.append(e1.kernel_ms(tk))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ach = abytes / (float(np.sum(kms)) * 1e-3) / 1e9
    out["resident_coalesce1"] = {"value": bases / wall / 1e6, "unit": "Mbases/s", "kernel_ms_per_batch": float(np.mean(kms)), "achieved": ach,
                                 "frac": ach / HBM_PEAK_GBS, "launches": e1.launch_counts(),
                                 "what": "the same %d steps, every step its own launch and waited for (mm_freq_opts_t.coalesce = 1)" % args.steps}
    e1.close()
    # ---- host batches through mm_freq_submit
    eh = minimod_amd.FreqEngine(wl["mods"], contig, coalesce=args.coalesce, **common)

    def host_pass():
        groups, kms = [], []
        for b in steps:
            tk = eh.submit(host_batches[b])
            if not groups or groups[-1] != tk:
                groups.append(tk)
            if len(groups) > 2:
                g = groups.pop(0)
                eh.wait(g)
                kms.append(eh.kernel_ms(g))
        for g in groups:
            eh.wait(g)
            kms.append(eh.kernel_ms(g))
        return kms
    host_pass()          # staging areas allocated, instantiations loaded
    eh.reset()
    walls, kmss = [], []
    lc0 = eh.launch_counts()
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = host_pass()
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        kmss.append(k)
        eh.reset()
    lc1 = eh.launch_counts()
    i = int(np.argsort(walls)[1])
    ach = abytes / (float(np.sum(kmss[i])) * 1e-3) / 1e9
    hbytes = int(sum(host_batches[b][k].nbytes for b in steps for k in ("reads", "cigar", "seq", "mm", "ml")))
    out["host_path"] = {"value": bases / walls[i] / 1e6, "unit": "Mbases/s", "wall_ms": walls[i] * 1e3, "h2d_bytes": hbytes,
                        "h2d_GBps_incl_kernels": hbytes / walls[i] / 1e9, "kernel_ms_per_batch": float(np.sum(kmss[i])) / args.steps,
                        "achieved_kernels_only": ach, "frac_kernels_only": ach / HBM_PEAK_GBS,
                        "launches_per_pass": (lc1["launches"] - lc0["launches"]) // 3, "stream_launches_per_pass": (lc1["stream_launches"] - lc0["stream_launches"]) // 3,
                        "what": "the same %d steps as HOST batches (pageable numpy memory) through mm_freq_submit with coalesce = %d: copied into a staging "
                                "area in HBM one behind the other, launched together; median of 3 passes, wall clock with the PCIe copies; "
                                "never the line's `value`" % (args.steps, args.coalesce)}
    eh.close()
    return out


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

    from minimod_amd import engine as _E
    dev_structs = [_E.batch_struct(db, device=True) for db in dev_batches]

    def run_steps(n, first_step=0):
        tickets, bases, kms, abytes, rows = [], 0, [], 0, 0
        for s in range(n):
            bi = (first_step + s) % n_batches
            tk = eng.submit_device(dev_structs[bi], stream)
            if not tickets or tickets[-1] != tk:   # (consecutive windows may share a launch and its ticket: --coalesce)
                tickets.append(tk)
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
        v_traffic, v_traffic_src = committed_traffic("view", 1.0)   # per step, like this line's `achieved`
        result = {
            "metric": "minimod view Mbases/sec", "value": total_bases / elapsed / 1e6, "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "C2 batches through view: %d ONT-shape reads per GPU, -c m[CG], -K %d, batches resident in HBM, "
                                   "ordered rows left in HBM" % (args.reads, args.batch),
                       "rows_per_step": rows / args.steps, "sharding": "interval per GPU, no exchange" if world > 1 else "single GPU"},
            # (round 5: `achieved` from the HIP events around the step's launches -- stream kernel + the three ordering kernels --, like the freq
            # line's and as the contract words it; rounds 1 - 4 divided by the wall clock of a step, which is kept beside it: frac_wall_clock)
            "roofline": {"bound": "hbm", "achieved": abytes / (float(np.sum(kms)) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": abytes / (float(np.sum(kms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_wall_clock": (abytes / args.steps) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": v_traffic, "traffic_source": v_traffic_src,
                         "kernel": "a step's launches = k_stream_reads<view> (or k_scan_reads + k_sum_tiles + k_call_tiles<view>) + k_view_offsets + k_view_scatter + "
                                   "k_view_sort + k_view_sort_big, HIP events around them on their stream (kernel_ms_mean: per launch of launches_per_run); "
                                   "frac_wall_clock: the same bytes over the wall clock of a step",
                         "launches": len(kms),
                         "kernel_ms_mean": mean_ms, "algorithmic_bytes_per_launch": abytes / args.steps},
            "gen_seconds": t_gen,
        }
        if args.dump:
            np.savez(args.dump, rows=eng.view(host_batches[0]))
            result["dump"] = {"file": args.dump, "batches": 1}
        if not args.no_cpu_baseline:
            from oracle import oracle as O
            cores = usable_cores()
            orc = O.Oracle([("m", "CG")], [0.8], [n for n, _ in plan["contigs"]])
            orc.set_view(True)
            for (n, _), rf in zip(plan["contigs"], ref):
                if rf is not None:
                    orc.add_contig(n, rf)
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


def cpu_baseline(args, wl, host_batches, plan, refs):
    """The oracle (bit-exact CPU restatement) timed on this host's cores over a bounded sample of the same batches."""
    from oracle import oracle as O
    cores = usable_cores()
    orc = O.Oracle([(c, x) for c, x, _ in wl["mods"]], [t for _, _, t in wl["mods"]], [n for n, _ in plan["contigs"]], **wl["eng"])
    for (n, _), rf in zip(plan["contigs"], refs):
        if rf is not None:
            orc.add_contig(n, rf)
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
