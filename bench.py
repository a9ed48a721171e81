#!/usr/bin/env python3
"""bench.py -- canonical-Huffman encode+decode throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic buffer per rank: histogram -> one-wave code
build -> bit-length scan -> bit-pack with header (encode), then table decode of the result.  The input is
resident in HBM before the timed region.  N=1: BASELINE config 2 (256 MiB uniform-random bytes).
N>1: every rank holds a 4 GiB shard of one N x 4 GiB stream (weak scaling; N = 8 is BASELINE config 4: 32 GiB);
one global code via an RCCL all-reduce of the 256-bin histogram and an all-gather of the per-rank bit totals.

Rank 0 prints ONE JSON line.  `value` = input GB (1e9 B) pushed through encode+decode per second by the
whole job.  `roofline` prices the dominant kernel against HBM peak; `cpu_baseline` is the reference's
own code (oracle/_ref, built from /root/reference in the build container) timed on this host, pinned to one
core.  `configs` (N=1) holds the other single-GPU BASELINE configs at their full 4 GiB size.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mib", type=int, default=None, help="input MiB per GPU (default: 256 at N=1 = config 2, 4096 at N>1 = config 4's shard)")
    ap.add_argument("--kind", default="uniform", choices=["uniform", "zipf", "sym16"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 control flow with several ranks on ONE GPU (collectives staged via host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the 4 GiB configs block (N=1)")
    args = ap.parse_args()
    if args.mib is None:
        args.mib = 256 if int(os.environ.get("WORLD_SIZE", "1")) == 1 else 4096
    return args


def cpu_baseline(data_host, kind):
    """the reference path on this host's CPU, 1 thread pinned to one core (the reference is single-threaded)."""
    import numpy as np

    from oracle import oracle as orc

    n = data_host.size
    cpu = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        core = sorted(os.sched_getaffinity(0))[-1]
    except (AttributeError, OSError):
        core = None
    if orc.have_ref():
        d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        fin, fout, fde = (os.path.join(d, "ghf_bench_%d.%s" % (os.getpid(), e)) for e in ("bin", "crs2", "de"))
        try:
            data_host.tofile(fin)
            t = json.loads(orc.ref_run(["b", fin, fout, fde], timeout=600, cpu=core))
            ok = os.path.getsize(fde) == n
        finally:
            for f in (fin, fout, fde):
                if os.path.exists(f):
                    os.remove(f)
        enc, dec = t["encode_total_s"], t["decode_bitserial_s"]
        return {"value": n / (enc + dec) / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference",
                "sample": "%d MiB %s (the whole N=1 workload), file-to-file in %s" % (n >> 20, kind, d),
                "encode_GBps": n / enc / 1e9, "decode_GBps": n / dec / 1e9, "histogram_GBps": n / t["histogram_s"] / 1e9,
                "round_trip_ok": bool(ok), "cpu": cpu, "host_cores": os.cpu_count(), "pinned_core": core}
    m = min(n, 64 << 20)
    sample = np.ascontiguousarray(data_host[:m])
    if core is not None:
        os.sched_setaffinity(0, {core})
    t0 = time.perf_counter()
    crs = orc.compress(sample)
    t1 = time.perf_counter()
    back = orc.decompress(crs, cap=m + 8)
    t2 = time.perf_counter()
    return {"value": m / (t2 - t0) / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "first %d MiB of the N=1 workload, in memory" % (m >> 20), "encode_GBps": m / (t1 - t0) / 1e9,
            "decode_GBps": m / (t2 - t1) / 1e9, "round_trip_ok": bool(np.array_equal(back, sample)), "cpu": cpu,
            "host_cores": os.cpu_count(), "pinned_core": core}


def copy_probe(torch, a, b, reps=5):
    """what a plain device-to-device copy of the same bytes reaches on this box (context for the roofline)"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.copy_(a)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * a.numel() * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def full_size_config(torch, ghf, synth, ctx, kind, mib, reps=5):
    """one single-GPU BASELINE config at full size, stage by stage, each stage timed with events around `reps` launches"""
    n = mib << 20
    d_in = synth.make(torch, kind, n, offset=0, device="cuda")
    out = ctx.empty_u8(ghf.compress_bound(n))
    dec = ctx.empty_u8(n)
    idx = ctx.index_alloc(n)
    hist = torch.empty(ghf.NSYM, dtype=torch.int64, device="cuda")
    code = ctx.new_code()
    end = torch.empty(2, dtype=torch.int64, device="cuda")
    ctx.use_current_stream()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_hist = timed(lambda: ctx.histogram(d_in, out=hist))
    t_code = timed(lambda: ctx.build_code(hist, code))
    ctx.histogram(d_in, out=hist)  # (the plan reuses the per-chunk histograms of the LAST histogram call on this input)
    t_plan = timed(lambda: ctx.encode_plan(d_in, code))
    t_emit = timed(lambda: ctx.encode_emit(d_in, code, out, flags=ghf.EMIT_LAST | ghf.EMIT_HEADER, index=idx, end=end))
    c = int(end[1].item())

    def dec_once():
        ctx.decode_prepare(code)
        ctx.decode(out, c, code, idx, d_out=dec)

    t_dec = timed(dec_once)
    ctx.sync()
    ok = bool((dec[:n] == d_in).all().item())
    probe = copy_probe(torch, d_in, dec, 3)
    ctx.index_free(idx)
    enc_ms = t_hist + t_plan + t_emit  # the streaming part of encode (the one-wave code build does not scale with n)
    res = {
        "workload": "%d MiB %s" % (mib, kind), "bytes": n, "compressed_bytes": c, "round_trip_ok": ok,
        "stage_ms": {"histogram": round(t_hist, 4), "build_code": round(t_code, 4), "plan": round(t_plan, 4), "emit": round(t_emit, 4),
                     "decode": round(t_dec, 4)},
        "encode": {"algorithmic_GBps": round((2 * n + c) / (enc_ms * 1e-3) / 1e9, 1), "read_side_GBps": round(2 * n / (enc_ms * 1e-3) / 1e9, 1),
                   "read_side_frac_of_8TBps": round(2 * n / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "ms": round(enc_ms, 4),
                   "ms_at_copy_rate": round((2 * n + c) / (probe * 1e9) * 1e3, 4)},
        "decode": {"algorithmic_GBps": round((c + n) / (t_dec * 1e-3) / 1e9, 1), "frac_of_8TBps": round((c + n) / (t_dec * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                   "output_GBps": round(n / (t_dec * 1e-3) / 1e9, 1), "ms_at_copy_rate": round((c + n) / (probe * 1e9) * 1e3, 4)},
        "copy_probe_GBps": round(probe, 1),
    }
    del d_in, out, dec
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    import torch

    import pkgload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.backend == "gloo":
        local_rank = 0  # rehearsal mode: every rank drives cuda:0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    pkg = pkgload.load()
    ghf = pkg.ghf
    from golden_huffman_amd import synth

    ctx = ghf.Context(local_rank)
    n = args.mib << 20
    d_in = synth.make(torch, args.kind, n, offset=rank * n, device="cuda")
    bound = ghf.compress_bound(n) if world == 1 else ghf.shard_bound(n)
    out = ctx.empty_u8(bound)
    dec = ctx.empty_u8(n)
    index = ctx.index_alloc(n)
    index.flags = 0 if rank == world - 1 else ghf.INDEX_NO_END_MARK  # only the last shard ends with the end mark
    torch.cuda.synchronize()

    names = ["histogram", "allreduce", "build_code", "header", "plan", "allgather", "emit", "decode"]
    acc_ms = {k: 0.0 for k in names}
    ev_log = []  # (name, start_event, end_event)

    # The one-wavefront code build (K2) is latency-bound (a strictly sequential heap on 1 of 256 CUs) and takes about
    # as long as all streaming kernels of a step together; the two collectives are latency-bound too.  So the steps
    # are software-pipelined, several in flight (DEPTH below): the main stream runs only the kernels that stream through HBM --
    # histogram of step i+2, emit and decode of step i -- while a side stream PER STEP IN FLIGHT runs, two steps ahead,
    # the histogram all-reduce, the code build, the chunk pricing (K4), the decode-table build and the offset
    # all-gather of step i+2 (one side stream for all steps would serialise the code builds of consecutive steps: the
    # 0.34 ms one-wave kernel then paces the whole pipeline).  Each step in flight has its own ghf context (= its own
    # workspace: the per-chunk histogram K1 leaves for K4, the chunk offsets K4 leaves for K5), which is also what
    # pipelining over DIFFERENT input buffers needs.  Every step does all of its work inside the timed region.
    # the streaming kernels run on a high-priority stream, the latency-bound helpers on normal ones: a helper's waves then
    # never sit in front of K7's on a CU (256 MiB: decode stage 0.150 -> 0.141 ms)
    main = torch.cuda.Stream(priority=-1)
    torch.cuda.set_stream(main)
    # Steps in flight.  Steady state needs three (K2 of step i+2 behind K5/K7 of step i); eight let the main stream count the
    # first seven inputs while the FIRST step's one-wave code build (0.37 ms, nothing to overlap it with at the start of a
    # run) is still going: at 20 timed steps that start-up is 0.04 ms per step, at 200 it does not show.
    DEPTH = int(os.environ.get("GHF_BENCH_DEPTH", "8"))  # (6 and 10 measure 5-10 % slower than 8 and 12, repeatably; not understood)
    ahead = DEPTH - 1
    NSIDE = int(os.environ.get("GHF_BENCH_NSIDE", "2"))
    sides = [torch.cuda.Stream(priority=0) for _ in range(NSIDE)]
    ctxs = [ctx] + [ghf.Context(local_rank) for _ in range(DEPTH - 1)]
    hists = [torch.empty(ghf.NSYM, dtype=torch.int64, device="cuda") for _ in range(DEPTH)]
    codes = [ctx.new_code() for _ in range(DEPTH)]
    t_total = [torch.empty(1, dtype=torch.int64, device="cuda") for _ in range(DEPTH)]
    t_totals = [torch.empty(max(world, 1), dtype=torch.int64, device="cuda") for _ in range(DEPTH)]
    t_start = [torch.empty(1, dtype=torch.int64, device="cuda") for _ in range(DEPTH)]
    ev_hist = [torch.cuda.Event() for _ in range(DEPTH)]
    ev_ready = [torch.cuda.Event() for _ in range(DEPTH)]
    t_end = torch.empty(2, dtype=torch.int64, device="cuda")
    t_nbytes = torch.empty(1, dtype=torch.int64, device="cuda")
    last_rank = rank == world - 1
    emit_flags = (ghf.EMIT_LAST if last_rank else 0) | (ghf.EMIT_REBASE if rank > 0 else ghf.EMIT_HEADER)  # rank 0 writes the header

    # N > 1: the two exchanges go through the C ABI (ghf_comm_*: RCCL called directly, queued on the step's side stream);
    # torch.distributed only carries the 128-byte ncclUniqueId.  One communicator per side stream.  If that cannot be set
    # up (or GHF_BENCH_COLLECTIVES=torch), torch.distributed's own collectives on the same stream are used instead.
    comms, coll_path = None, "none"
    if world > 1:
        coll_path = "torch.distributed (%s)" % args.backend
        if args.backend == "nccl" and os.environ.get("GHF_BENCH_COLLECTIVES", "cabi") == "cabi":
            try:
                ids = [ghf.comm_unique_id() if rank == 0 else None for _ in range(NSIDE)]
                dist.broadcast_object_list(ids, src=0)
                comms = [ctx.comm_init(ids[j], world, rank) for j in range(NSIDE)]
                coll_path = "C ABI: ghf_comm_allreduce_hist / ghf_comm_allgather_total (RCCL %s called directly)" % ghf.rccl_version()
            except Exception as e:
                comms = None
                coll_path += " [C-ABI communicator failed: %s]" % repr(e)[:120]

    def all_reduce_sum(t):
        if args.backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            t.copy_(c)

    def all_gather_1(out_t, t):
        if args.backend == "nccl":
            dist.all_gather_into_tensor(out_t, t)
        else:
            parts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(parts, t.cpu())
            out_t.copy_(torch.cat(parts))

    def timed(name, record, stream, fn):
        if not record:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r = fn()
        e1.record(stream)
        ev_log.append((name, e0, e1))
        return r

    def front(i, record):
        """step i up to the point where its emit can start: K1 on the main stream; all-reduce, K2/K3, K4, decode tables,
        all-gather on this step's side stream."""
        k = i % DEPTH
        cx, h, c, side = ctxs[k], hists[k], codes[k], sides[i % NSIDE]
        cx.use_stream(main)
        timed("histogram", record, main, lambda: cx.histogram(d_in, out=h))
        ev_hist[k].record(main)
        side.wait_event(ev_hist[k])
        cx.use_stream(side)
        if world > 1:
            if comms:
                timed("allreduce", record, side, lambda: cx.comm_allreduce_hist(comms[i % NSIDE], h))
            else:
                with torch.cuda.stream(side):
                    timed("allreduce", record, side, lambda: all_reduce_sum(h[:256]))
        timed("build_code", record, side, lambda: cx.build_code(h, c))
        timed("plan", record, side, lambda: cx.encode_plan(d_in, c, total=t_total[k]))
        cx.decode_prepare(c)  # the decode tables of this code: one tiny kernel less on the main stream
        if world > 1:
            def gather():
                if comms:
                    cx.comm_allgather_total(comms[i % NSIDE], t_total[k], t_totals[k])
                else:
                    with torch.cuda.stream(side):
                        all_gather_1(t_totals[k], t_total[k])
                cx.shard_start_bit(c, t_totals[k], world, rank, out=t_start[k])
            timed("allgather", record, side, gather)
        ev_ready[k].record(side)

    def run(K, record):
        end = None
        rec_of = lambda j: record and (j % 4 == 1 or K <= 4)  # events on every 4th step: keeps the host ahead of the GPU
        for j in range(min(ahead, K)):
            front(j, rec_of(j))
        for i in range(K):
            k = i % DEPTH
            cx, c = ctxs[k], codes[k]
            if i + ahead < K:
                front(i + ahead, rec_of(i + ahead))  # (uses the events of slot (i + 2) % 3, not this step's)
            main.wait_event(ev_ready[k])
            cx.use_stream(main)
            start_bit = t_start[k] if world > 1 else None
            end = timed("emit", rec_of(i), main, lambda: cx.encode_emit(d_in, c, out, start_bit=start_bit, flags=emit_flags, index=index, end=t_end))
            timed("decode", rec_of(i), main, lambda: cx.decode(out, bound, c, index, d_out=dec, nbytes=t_nbytes))
        return end

    end = run(max(args.warmup, 1), False)
    torch.cuda.synchronize()
    for cx in ctxs:
        cx.sync()
    if not args.no_verify:
        assert bool((dec[:n] == d_in).all().item()), "round trip mismatch"
    comp_bytes = int(end[1].item())

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for cx in ctxs:
        cx.sync()  # raises if any stage latched an error
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    counts = {k: 0 for k in names}
    for name, e0, e1 in ev_log:
        acc_ms[name] += e0.elapsed_time(e1)
        counts[name] += 1
    stage_ms = {k: (acc_ms[k] / counts[k] if counts[k] else 0.0) for k in names}

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_in = n * world
        value = total_in * args.steps / elapsed / 1e9
        enc_ms = sum(stage_ms[k] for k in names[:7])  # un-overlapped sum of the encode stages (latency of one buffer)
        # roofline of the dominant kernel (algorithmic bytes): emit reads N and writes the body,
        # decode reads C and writes N, histogram reads N
        cand = {"k_emit": (n + comp_bytes, stage_ms["emit"]), "k_decode": (comp_bytes + n, stage_ms["decode"]),
                "k_histogram": (n, stage_ms["histogram"])}
        dom = max(cand, key=lambda k: cand[k][1])
        ach = cand[dom][0] / (cand[dom][1] * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("%s_%s_%dMiB" % (dom, args.kind, args.mib))
            except Exception:
                traffic = None
        copy_gbps = copy_probe(torch, d_in, dec)
        cfg4 = world > 1 and args.kind == "uniform" and args.mib == 4096
        res = {
            "metric": "encode+decode GB/s (input bytes)", "value": round(value, 3), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d MiB %s bytes per GPU, encode (.crs2 bit-exact with the reference) + decode" % (args.mib, args.kind),
                       "baseline_config": "configs[1]" if (world == 1 and args.kind == "uniform" and args.mib == 256) else ("configs[3]" if cfg4 else "configs[3]-style shard"),
                       "bytes_per_gpu": n, "compressed_bytes_per_gpu": comp_bytes, "parallelism": "shard%d" % world,
                       "collectives": "none" if world == 1 else "all_reduce(256 x i64) + all_gather(1 x i64) per step",
                       "world_size": world, "collective_path": coll_path, "backend": ("none" if world == 1 else ("rccl (torch.distributed nccl)" if args.backend == "nccl" else "gloo (rehearsal)")),
                       "pipeline": "steps software-pipelined, up to %d in flight (one ghf context each, two side streams): main stream = histogram of step i+%d, emit + decode of step i; side streams, ahead of the main one = histogram all-reduce, one-wave code build, chunk pricing, decode tables, offset all-gather" % (DEPTH, DEPTH - 1)},
            "encode_GBps": round(n * world / (enc_ms * 1e-3) / 1e9, 3), "decode_GBps": round(n * world / (stage_ms["decode"] * 1e-3) / 1e9, 3),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": cand[dom][0], "avg_launch_ms": round(cand[dom][1], 4),
                         "copy_probe_GBps": round(copy_gbps, 1),
                         "all_kernels": {k: {"achieved": round(v[0] / (v[1] * 1e-3) / 1e9, 1), "frac": round(v[0] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
                                         for k, v in cand.items() if v[1] > 0}},
            "library": ghf.lib_identity(),
        }
        if world == 1:
            # the decode path a stream WITHOUT side-car takes (a .crs2 the reference wrote; what Decompressor<...>::decompress()
            # of the C++ host layer runs): K6 rebuilds the side-car on the GPU, then K7.  Host wall time: K6 looks at a
            # convergence word from the host a few times.
            cx = ctxs[0]
            cx.use_stream(main)
            cx.decode(out, comp_bytes, codes[0], None, d_out=dec, cap=n, nbytes=t_nbytes)
            torch.cuda.synchronize()
            reps = 5
            tf0 = time.perf_counter()
            for _ in range(reps):
                cx.decode(out, comp_bytes, codes[0], None, d_out=dec, cap=n, nbytes=t_nbytes)
            torch.cuda.synchronize()
            tf = (time.perf_counter() - tf0) / reps
            cx.sync()
            ok_f = int(t_nbytes.item()) == n and (args.no_verify or bool((dec[:n] == d_in).all().item()))
            res["decode_foreign"] = {"GBps": round(n / tf / 1e9, 3), "ms": round(tf * 1e3, 4), "round_trip_ok": bool(ok_f),
                                     "vs_indexed_decode": round(tf * 1e3 / stage_ms["decode"], 2) if stage_ms["decode"] else None,
                                     "what": "ghf_decode(index = NULL): K6 side-car reconstruction + K7, host wall time per stream"}
        if world > 1 and args.backend == "nccl":
            try:
                res["config"]["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                pass
        if world == 1 and not args.no_configs:
            # the other single-GPU BASELINE configs at full size: 4 GiB uniform (north star: encode read side), configs[2]
            # (4 GiB Zipf encode) and configs[4] (4 GiB 16-symbol decode); every number has its stage times beside it
            del out, dec
            torch.cuda.empty_cache()
            blk = {}
            for key, kind in (("uniform_4GiB", "uniform"), ("configs[2]_zipf_4GiB", "zipf"), ("configs[4]_sym16_4GiB", "sym16")):
                try:
                    blk[key] = full_size_config(torch, ghf, synth, ctx, kind, 4096)
                except Exception as e:  # (e.g. not enough free HBM on a shared box)
                    blk[key] = {"error": repr(e)[:200]}
            res["configs"] = blk
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(d_in.cpu().numpy(), args.kind)
        print(json.dumps(res), flush=True)
    ctx.index_free(index)
    for cm in comms or []:
        ctx.comm_destroy(cm)
    for cx in reversed(ctxs):
        cx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
