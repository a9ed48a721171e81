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

Steps in flight work on DIFFERENT buffers: GHF_BENCH_SETS (default 3) sets of {input, output, decoded, side-car}, each with
its own synthetic data, rotate with the step number.  A watchdog thread ends the run with the name of the stuck stage when
no stage completes for GHF_BENCH_STEP_TIMEOUT seconds (default 180) -- a collective that never returns must not sit there
until the driver's limit.

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


def _time_gbps(torch, fn, nbytes, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def copy_probe(torch, ctx, a, b, reps=5):
    """what a kernel that only MOVES the same bytes reaches on this box (context for the roofline): the library's own streaming
    copy kernel (ghf_copy_d2d: 16 B per lane, four loads in flight, non-temporal hints -- the access shape of K1/K5/K7), and
    torch's copy_ beside it.  GB/s of read + write."""
    ctx.use_current_stream()
    n = a.numel() & ~15
    return {"copy_probe_GBps": round(_time_gbps(torch, lambda: ctx.copy_d2d(b, a, n, non_temporal=True), 2.0 * n, reps), 1),
            "copy_probe_plain_GBps": round(_time_gbps(torch, lambda: ctx.copy_d2d(b, a, n, non_temporal=False), 2.0 * n, reps), 1),
            "torch_copy_GBps": round(_time_gbps(torch, lambda: b.copy_(a), 2.0 * a.numel(), reps), 1)}


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
    probes = copy_probe(torch, ctx, d_in, dec, 3)
    probe = probes["copy_probe_GBps"]
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
        "single_buffer_ms": round(t_hist + t_code + t_plan + t_emit + t_dec, 4),
    }
    res.update(probes)
    del d_in, out, dec
    torch.cuda.empty_cache()
    return res


class Watchdog:
    """Ends the process with the name of the stuck stage when nothing completes for `timeout` seconds.  The main thread
    notes what it is about to queue (`enter`) and leaves an event behind every stage (`mark`); a daemon thread looks at
    the oldest event that has not completed yet.  It never touches the GPU except through event.query()."""

    def __init__(self, timeout, rank):
        import threading

        self.timeout, self.rank = timeout, rank
        self.pending = []  # (label, event, t_recorded), oldest first
        self.host = ("start", time.monotonic())
        self.lock = threading.Lock()
        self.stop = False
        self.t = threading.Thread(target=self._run, daemon=True)
        if timeout > 0:
            self.t.start()

    def enter(self, label):
        self.host = (label, time.monotonic())

    def mark(self, label, event):
        if self.timeout > 0:
            with self.lock:
                self.pending.append((label, event, time.monotonic()))

    def close(self):
        self.stop = True

    def _fire(self, why):
        sys.stderr.write("bench.py: rank %d: no progress for %.0f s -- %s\n" % (self.rank, self.timeout, why))
        sys.stderr.flush()
        os._exit(3)  # (exit, never exec: the GPU is initialised)

    def _run(self):
        while not self.stop:
            time.sleep(0.25)
            now = time.monotonic()
            with self.lock:
                while self.pending and self.pending[0][1].query():
                    self.pending.pop(0)
                oldest = self.pending[0] if self.pending else None
            if oldest is not None and now - oldest[2] > self.timeout:
                self._fire("the GPU has not finished stage '%s' (queued %.0f s ago); the host is at '%s'" % (oldest[0], now - oldest[2], self.host[0]))
            if oldest is None and now - self.host[1] > self.timeout and self.host[0] not in ("done", "start"):
                self._fire("the host is stuck in '%s'" % self.host[0])


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
        import datetime

        import torch.distributed as dist

        tmo = datetime.timedelta(seconds=int(os.environ.get("GHF_BENCH_PG_TIMEOUT", "300")))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group("gloo", timeout=tmo)
    pkg = pkgload.load()
    ghf = pkg.ghf
    from golden_huffman_amd import synth

    wd = Watchdog(float(os.environ.get("GHF_BENCH_STEP_TIMEOUT", "180")), rank)
    hang_at = os.environ.get("GHF_BENCH_INJECT_HANG", "")  # "<rank>:<step>": that rank never joins that step's all-reduce (tests)
    ctx = ghf.Context(local_rank)
    n = args.mib << 20
    bound = ghf.compress_bound(n) if world == 1 else ghf.shard_bound(n)
    # Buffer sets: steps in flight work on different inputs and write different outputs.  Set s of rank r is the byte range
    # [(s * world + r) * n, +n) of one long synthetic stream, so that (for every s) the ranks' shards are consecutive.
    NMAIN_DEFAULT = "2" if world == 1 else "1"  # (N > 1: one main stream -- collectives in strict step order, as rehearsed)
    NSETS = max(1, int(os.environ.get("GHF_BENCH_SETS", "3" if os.environ.get("GHF_BENCH_MAINS", NMAIN_DEFAULT) == "1" else "4")))

    class BufSet:
        pass

    sets = []
    for si in range(NSETS):
        b = BufSet()
        b.d_in = synth.make(torch, args.kind, n, offset=(si * world + rank) * n, device="cuda")
        b.out = ctx.empty_u8(bound)
        b.bound = bound
        b.dec = ctx.empty_u8(n)
        b.index = ctx.index_alloc(n)
        b.index.flags = 0 if rank == world - 1 else ghf.INDEX_NO_END_MARK  # only the last shard ends with the end mark
        b.end = torch.empty(2, dtype=torch.int64, device="cuda")
        sets.append(b)
    torch.cuda.synchronize()

    names = ["histogram", "allreduce", "build_code", "header", "plan", "allgather", "emit", "decode"]

    # The one-wavefront code build (K2) is latency-bound (a strictly sequential heap on 1 of 256 CUs) and the two
    # collectives are latency-bound too.  So the steps are software-pipelined, several in flight (DEPTH below): the main
    # stream runs only the kernels that stream through HBM -- histogram of step i+DEPTH-1, emit and decode of step i --
    # while side streams run, ahead of it, the histogram all-reduce, the code build, the chunk pricing (K4), the
    # decode-table build and the offset all-gather.  Each step in flight has its own ghf context (= its own workspace:
    # the per-chunk histogram K1 leaves for K4, the chunk offsets K4 leaves for K5) and works on buffer set i % NSETS.
    # Every step does all of its work inside the timed region.
    # The streaming kernels run on a high-priority stream, the latency-bound helpers on normal ones: a helper's waves then
    # never sit in front of K7's on a CU (256 MiB: decode stage 0.150 -> 0.141 ms).
    main = torch.cuda.Stream(priority=-1)
    torch.cuda.set_stream(main)
    # GHF_BENCH_K1_STREAM=1: the histograms get a high-priority stream of their own, so that K1 of a later step fills the
    # ramp-up and the tail of K5 / K7 of the current one (all three only stream through HBM)
    pre = torch.cuda.Stream(priority=-1) if os.environ.get("GHF_BENCH_K1_STREAM", "0") == "1" else main
    # GHF_BENCH_MAINS=2 (the default at N = 1): even and odd steps pack and decode on two high-priority streams, so that one
    # step's K5 fills the ramp and the tail of the other's K7 (+6 % throughput at 256 MiB, profiles/r03/experiments/
    # bench_b_m2_k0.json).  Needs an even number of buffer sets and of steps in flight (a set / a context slot is then always
    # used from the same stream).  The events of the timed region then time kernels that SHARE the GPU (`stage_ms`); the
    # kernels' own durations -- what `roofline` prices -- come from a separate pass over the same buffers with nothing
    # else running (`stage_ms_alone`, measured right behind the timed region).
    NMAIN = int(os.environ.get("GHF_BENCH_MAINS", NMAIN_DEFAULT))
    mains = [main] + [torch.cuda.Stream(priority=-1) for _ in range(NMAIN - 1)]
    # Steps in flight.  Steady state needs three; more let the main stream count the first inputs while the FIRST step's
    # one-wave code build (nothing to overlap it with at the start of a run) is still going.
    FRONT_FIRST = os.environ.get("GHF_BENCH_FRONT_FIRST", "1") == "1"
    RAMP0 = int(os.environ.get("GHF_BENCH_RAMP0", "3"))
    DEPTH = int(os.environ.get("GHF_BENCH_DEPTH", "8"))
    ahead = DEPTH - 1
    # Side streams.  N = 1: four.  A step's side chain -- code build 0.26 ms alone and 0.35 between the streaming kernels,
    # chunk pricing, decode tables: 0.5 ms -- has to keep up with a main pipeline of 0.29 ms per step: one side stream paces
    # the run outright, two still did (865 GB/s; three 900, four 922, eight 920).  N > 1: ONE, and one communicator -- collectives are then issued in strict step order on one stream on
    # every rank, which is the only order RCCL guarantees to match up across ranks (two communicators' kernels may be
    # launched in different orders on different ranks: a deadlock that needs all 8 GPUs to show).  GHF_BENCH_NSIDE
    # overrides (one communicator per side stream).
    NSIDE = int(os.environ.get("GHF_BENCH_NSIDE", "4" if world == 1 else "1"))
    sides = [torch.cuda.Stream(priority=0) for _ in range(NSIDE)]
    ctxs = [ctx] + [ghf.Context(local_rank) for _ in range(DEPTH - 1)]
    hists = [torch.empty(ghf.NSYM, dtype=torch.int64, device="cuda") for _ in range(DEPTH)]
    codes = [ctx.new_code() for _ in range(DEPTH)]
    t_total = [torch.empty(1, dtype=torch.int64, device="cuda") for _ in range(DEPTH)]
    t_totals = [torch.empty(max(world, 1), dtype=torch.int64, device="cuda") for _ in range(DEPTH)]
    t_start = [torch.empty(1, dtype=torch.int64, device="cuda") for _ in range(DEPTH)]
    ev_hist = [torch.cuda.Event() for _ in range(DEPTH)]
    ev_ready = [torch.cuda.Event() for _ in range(DEPTH)]
    ev_done = [torch.cuda.Event() for _ in range(DEPTH)]  # the slot's step has decoded: its context, code and tables may be rewritten
    t_nbytes = torch.empty(1, dtype=torch.int64, device="cuda")
    last_rank = rank == world - 1
    sharded_flags = (ghf.EMIT_LAST if last_rank else 0) | (ghf.EMIT_REBASE if rank > 0 else ghf.EMIT_HEADER)  # rank 0 writes the header
    local_flags = ghf.EMIT_LAST | ghf.EMIT_HEADER

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

    class Run:
        """one pipelined run of K steps.  sharded = False: every rank for itself (no collectives, the shard is a whole stream)."""

        def __init__(self, sharded, comms):
            self.sharded, self.comms = sharded, comms
            self.ev_log = []  # (name, start_event, end_event)
            self.host_front_s, self.host_fronts = 0.0, 0  # host time spent enqueueing fronts (diagnostic: must stay well below the step time)

        def timed(self, name, step, record, stream, fn):
            wd.enter("%s of step %d" % (name, step))
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                r = fn()
                e1.record(stream)
                self.ev_log.append((name, e0, e1))
            else:
                r = fn()
                e1 = None
            if name in ("allreduce", "allgather", "decode"):  # what the watchdog looks at: the collectives and every step's end
                if e1 is None:
                    e1 = torch.cuda.Event()
                    e1.record(stream)
                wd.mark("%s of step %d" % (name, step), e1)
            return r

        def front(self, i, record):
            """step i up to the point where its emit can start: K1 on the main stream; all-reduce, K2/K3, K4, decode
            tables, all-gather on a side stream."""
            th0 = time.perf_counter()
            try:
                self._front(i, record)
            finally:
                self.host_front_s += time.perf_counter() - th0
                self.host_fronts += 1

        def _front(self, i, record):
            k = i % DEPTH
            cx, h, c, side, b = ctxs[k], hists[k], codes[k], sides[i % NSIDE], sets[i % NSETS]
            comm = self.comms[i % NSIDE] if self.comms else None
            if i >= DEPTH and (pre is not main or NMAIN > 1):
                # this slot's previous step is through: K1 overwrites the per-chunk counts its K4 read, and the side-stream
                # kernels behind K1 rewrite the code and the decode tables its K5 / K7 read.  (With ONE main stream that carries
                # K1 too, stream order says so already: the previous step's decode was queued on it before this K1.)
                pre.wait_event(ev_done[k])
            cx.use_stream(pre)
            self.timed("histogram", i, record, pre, lambda: cx.histogram(b.d_in, out=h))
            ev_hist[k].record(pre)
            side.wait_event(ev_hist[k])
            cx.use_stream(side)
            if self.sharded:
                if hang_at == "%d:%d" % (rank, i):
                    wd.enter("allreduce of step %d (GHF_BENCH_INJECT_HANG: this rank stays away)" % i)
                    time.sleep(1e6)
                if comm:
                    self.timed("allreduce", i, record, side, lambda: cx.comm_allreduce_hist(comm, h))
                else:
                    with torch.cuda.stream(side):
                        self.timed("allreduce", i, record, side, lambda: all_reduce_sum(h[:256]))
            self.timed("build_code", i, record, side, lambda: cx.build_code(h, c))
            self.timed("plan", i, record, side, lambda: cx.encode_plan(b.d_in, c, total=t_total[k]))
            cx.decode_prepare(c)  # the decode tables of this code: one tiny kernel less on the main stream
            if self.sharded:
                def gather():
                    if comm:
                        cx.comm_allgather_total(comm, t_total[k], t_totals[k])
                    else:
                        with torch.cuda.stream(side):
                            all_gather_1(t_totals[k], t_total[k])
                    cx.shard_start_bit(c, t_totals[k], world, rank, out=t_start[k])
                self.timed("allgather", i, record, side, gather)
            ev_ready[k].record(side)

        def run(self, K, record):
            rec_of = lambda j: record and (j % 4 == 1 or K <= 4)  # events on every 4th step: keeps the host ahead of the GPU
            # The look-ahead is RAMPED: three steps' fronts first (their enqueueing takes the host as long as the first code
            # build takes the GPU), then the first emit / decode, then two fronts per step until the host is `ahead` steps
            # in front.  Enqueueing all `ahead` fronts first left the GPU with nothing but eight histograms for the first
            # millisecond of a run (0.1 ms of host work per front: scratch/trace_timeline.py) -- 0.5 ms per run, 8 % of a
            # 20-step one.
            nf = 0  # fronts enqueued
            while nf < min(ahead, RAMP0, K):
                self.front(nf, rec_of(nf))
                nf += 1
            for i in range(K):
                k = i % DEPTH
                cx, c, b = ctxs[k], codes[k], sets[i % NSETS]
                while nf <= i:  # (only with GHF_BENCH_DEPTH=1: no look-ahead at all)
                    self.front(nf, rec_of(nf))
                    nf += 1
                if FRONT_FIRST:
                    for _ in range(2):
                        if nf < min(K, i + 1 + ahead):
                            self.front(nf, rec_of(nf))
                            nf += 1
                mstream = mains[i % NMAIN]
                mstream.wait_event(ev_ready[k])
                cx.use_stream(mstream)
                start_bit = t_start[k] if self.sharded else None
                flags = sharded_flags if self.sharded else local_flags
                b.index.flags = (0 if last_rank else ghf.INDEX_NO_END_MARK) if self.sharded else 0
                self.timed("emit", i, rec_of(i), mstream, lambda: cx.encode_emit(b.d_in, c, b.out, start_bit=start_bit, flags=flags, index=b.index, end=b.end))
                self.timed("decode", i, rec_of(i), mstream, lambda: cx.decode(b.out, b.bound, c, b.index, d_out=b.dec, nbytes=t_nbytes))
                ev_done[k].record(mstream)
                for _ in range(0 if FRONT_FIRST else 2):  # (behind the step's own kernels: its emit must not wait for the host to enqueue other steps' fronts)
                    if nf < min(K, i + 1 + ahead):
                        self.front(nf, rec_of(nf))
                        nf += 1

        def measure(self, K, barrier):
            """time exactly K steps: barrier + synchronize on both sides"""
            if barrier:
                wd.enter("barrier before the timed region")
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.run(K, True)
            wd.enter("synchronize after the timed region")
            torch.cuda.synchronize()
            if barrier:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            for cx in ctxs:
                cx.sync()  # raises if any stage latched an error
            return elapsed

        def stage_ms(self):
            acc, cnt = {k: 0.0 for k in names}, {k: 0 for k in names}
            for name, e0, e1 in self.ev_log:
                acc[name] += e0.elapsed_time(e1)
                cnt[name] += 1
            return {k: (acc[k] / cnt[k] if cnt[k] else 0.0) for k in names}

    def verify_sets(what):
        torch.cuda.synchronize()
        for cx in ctxs:
            cx.sync()
        if not args.no_verify:
            for si, b in enumerate(sets):
                assert bool((b.dec[:n] == b.d_in).all().item()), "round trip mismatch (%s, buffer set %d)" % (what, si)

    extra = {}
    # ---- N > 1, before any communicator exists: the SAME shard on this GPU alone, same pipeline, no collectives.  The
    # speed-up of the N-GPU run is quoted against this number (same bytes per GPU), never against the 256 MiB N = 1 line.
    if world > 1:
        solo = Run(False, None)
        solo.run(NSETS, False)
        verify_sets("same shard, this GPU alone")
        ks = max(4, min(args.steps, 10))
        t_solo = solo.measure(ks, False)
        extra["same_shard_1gpu_ms_per_step"] = round(t_solo * 1e3 / ks, 4)
        extra["same_shard_1gpu_steps"] = ks
        extra["same_shard_1gpu_GBps"] = round(n * ks / t_solo / 1e9, 3)

    # N > 1: the two exchanges go through the C ABI (ghf_comm_*: RCCL called directly, queued on the step's side stream);
    # torch.distributed only carries the 128-byte ncclUniqueId.  If that cannot be set up (or GHF_BENCH_COLLECTIVES=torch),
    # torch.distributed's own collectives on the same stream are used instead.
    comms, coll_path = None, "none"
    if world > 1:
        coll_path = "torch.distributed (%s)" % args.backend
        if args.backend == "nccl" and os.environ.get("GHF_BENCH_COLLECTIVES", "cabi") == "cabi":
            try:
                wd.enter("communicator set-up")
                ids = [ghf.comm_unique_id() if rank == 0 else None for _ in range(NSIDE)]
                dist.broadcast_object_list(ids, src=0)
                comms = [ctx.comm_init(ids[j], world, rank) for j in range(NSIDE)]
                coll_path = "C ABI: ghf_comm_allreduce_hist / ghf_comm_allgather_total (RCCL %s called directly)" % ghf.rccl_version()
            except Exception as e:
                comms = None
                coll_path += " [C-ABI communicator failed: %s]" % repr(e)[:120]

    job = Run(world > 1, comms)
    job.run(NSETS, False)  # every buffer set once, verified (not part of --warmup)
    verify_sets("sharded pipeline" if world > 1 else "pipeline")
    if world > 1 and NSETS <= DEPTH:
        # The first pass ran into buffers of the static bound (4 n: what fits ANY code).  The all-gathered bit totals of that
        # pass say what every shard really takes: from here on the outputs are exactly that large (ghf_shard_bytes; K5
        # latches GHF_E_CAP if it ever is not enough) -- 4.3 GB per buffer instead of 17.2 at config 4.
        wd.enter("exact shard outputs")
        exact = []
        for si, b in enumerate(sets):  # (step si of the pass above used context slot si and buffer set si)
            nb = ctxs[si].shard_bytes(codes[si], t_totals[si], world, rank)
            b.out = None
            exact.append(nb)
        torch.cuda.empty_cache()
        for b, nb in zip(sets, exact):
            b.out = ctx.empty_u8(nb)
            b.bound = nb
        extra["shard_output_bytes"] = {"per_buffer_set": exact, "static_bound": bound, "sized_by": "ghf_shard_bytes(all-gathered totals) after the first pass"}
        job.run(NSETS, False)
        verify_sets("sharded pipeline, exact outputs")
    if world > 1 and comms:
        # one step through the ONE-CALL entry point (ghf_encode_sharded: K1 -> all-reduce -> K2/K3 -> K4 -> all-gather -> K5 on
        # one stream): its bytes must equal what the staged calls above wrote for the same shard, and decode back
        wd.enter("ghf_encode_sharded (verified step)")
        b = sets[0]
        ctx.use_stream(main)
        alt, aidx = ctx.empty_u8(bound), ctx.index_alloc(n)
        r = ctx.encode_sharded(comms[0], b.d_in, d_out=alt, index=aidx)
        torch.cuda.synchronize()
        ctx.sync()
        nb = int(r["end"][1].item())
        same = nb == int(b.end[1].item()) and bool((alt[:nb] == b.out[:nb]).all().item())
        back, _ = ctx.decode(alt, bound, r["code"], aidx)
        ctx.sync()
        ok = same and bool((back[:n] == b.d_in).all().item())
        extra["encode_sharded_c_abi"] = {"bytes_equal_staged_pipeline": bool(same), "round_trip_ok": bool(ok), "shard_bytes": nb}
        assert ok or args.no_verify, "ghf_encode_sharded differs from the staged pipeline on rank %d" % rank
        ctx.index_free(aidx)
        del alt, back
    job.run(max(args.warmup, 1), False)
    verify_sets("warm-up")
    comp_bytes = int(sets[(max(args.warmup, 1) - 1) % NSETS].end[1].item())

    elapsed = job.measure(args.steps, world > 1)
    verify_sets("timed region")  # what the timed steps left in the buffers decodes back to the inputs
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage_ms = job.stage_ms()

    def alone_ms(reps=20):
        """every streaming kernel's OWN duration: one kernel at a time on the main stream, nothing else queued anywhere, HIP
        events on that stream around each launch, rotating over the buffer sets (the same bytes the timed region moved)"""
        torch.cuda.synchronize()
        acc = {"histogram": [], "build_code": [], "plan": [], "emit": [], "decode": []}
        cx = ctxs[0]
        cx.use_stream(main)
        flags = local_flags  # (N > 1: the shard as a whole stream -- the kernels' own time does not depend on the collectives)
        # N > 1: the sets' output buffers hold exactly this rank's part of the JOB's stream (ghf_shard_bytes); the shard as a
        # stream of its own (own header, own code) needs the static bound
        own_out = ctx.empty_u8(bound) if world > 1 else None
        for r in range(reps):
            b = sets[r % NSETS]
            b.index.flags = 0
            o_buf, o_cap = (own_out, bound) if world > 1 else (b.out, b.bound)

            def ev(name, fn):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                fn()
                e1.record(main)
                acc[name].append((e0, e1))

            ev("histogram", lambda: cx.histogram(b.d_in, out=hists[0]))
            ev("build_code", lambda: cx.build_code(hists[0], codes[0]))
            ev("plan", lambda: cx.encode_plan(b.d_in, codes[0], total=t_total[0]))
            ev("emit", lambda: cx.encode_emit(b.d_in, codes[0], o_buf, flags=flags, index=b.index, end=b.end))
            cx.decode_prepare(codes[0])  # (k_build_decode_tables: a side-stream kernel of the pipeline, not part of K7)
            ev("decode", lambda: cx.decode(o_buf, o_cap, codes[0], b.index, d_out=b.dec, nbytes=t_nbytes))
            torch.cuda.synchronize()  # one launch in flight at a time, and no host queueing effects in the events
        cx.sync()
        return {k: sum(a.elapsed_time(z) for a, z in v) / len(v) for k, v in acc.items()}

    wd.enter("per-kernel pass")
    stage_alone = alone_ms()
    verify_sets("per-kernel pass")
    wd.enter("done")

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_in = n * world
        value = total_in * args.steps / elapsed / 1e9
        enc_ms = sum(stage_alone.get(k, 0.0) for k in names[:7])  # un-overlapped sum of the encode stages
        # roofline of the dominant kernel (algorithmic bytes): emit reads N and writes the body,
        # decode reads C and writes N, histogram reads N
        cand = {"k_emit": (n + comp_bytes, stage_alone["emit"]), "k_decode": (comp_bytes + n, stage_alone["decode"]),
                "k_histogram": (n, stage_alone["histogram"])}
        dom = max(cand, key=lambda k: cand[k][1])
        ach = cand[dom][0] / (cand[dom][1] * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("%s_%s_%dMiB" % (dom, args.kind, args.mib))
            except Exception:
                traffic = None
        probes = copy_probe(torch, ctx, sets[0].d_in, sets[0].dec)
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
                       "buffer_sets": NSETS, "side_streams": NSIDE, "histogram_stream": "own" if pre is not main else "main", "communicators": len(comms) if comms else 0,
                       "pipeline": "steps software-pipelined, up to %d in flight (one ghf context each, %d side stream(s)), rotating over %d sets of {input, output, decoded, side-car} buffers: main stream = histogram of step i+%d, emit + decode of step i%s; side stream(s), ahead of the main one = histogram all-reduce, one-wave code build, chunk pricing, decode tables, offset all-gather.  stage_ms = events inside the timed region (kernels of different streams share the GPU); stage_ms_alone and roofline = a separate pass behind it, one kernel at a time" % (DEPTH, NSIDE, NSETS, DEPTH - 1, " (even and odd steps on two main streams)" if NMAIN > 1 else "")},
            "main_streams": NMAIN,
            "host_enqueue_ms_per_front": round(job.host_front_s * 1e3 / max(job.host_fronts, 1), 4),
            "encode_GBps": round(n * world / (enc_ms * 1e-3) / 1e9, 3), "decode_GBps": round(n * world / (stage_alone["decode"] * 1e-3) / 1e9, 3),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_alone": {k: round(v, 4) for k, v in stage_alone.items()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, builder-run, corrected as the guide prescribes; not measured in this run)" if traffic is not None else None,
                         "algorithmic_bytes_per_launch": cand[dom][0], "avg_launch_ms": round(cand[dom][1], 4),
                         "timed_by": "HIP events on the kernel's stream around each of 20 launches, one kernel on the GPU at a time, right behind the timed region (stage_ms_alone); the same kernel inside the timed region, sharing the GPU with the other stream's kernels: %.4f ms (stage_ms)" % stage_ms[{"k_emit": "emit", "k_decode": "decode", "k_histogram": "histogram"}[dom]],
                         "all_kernels": {k: {"achieved": round(v[0] / (v[1] * 1e-3) / 1e9, 1), "frac": round(v[0] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
                                         for k, v in cand.items() if v[1] > 0}},
            "library": ghf.lib_identity(),
        }
        res["roofline"].update(probes)
        res.update(extra)
        if "same_shard_1gpu_ms_per_step" in extra:
            res["speedup_vs_same_shard"] = round(world * extra["same_shard_1gpu_ms_per_step"] / ms_per_step, 3)
        if world == 1:
            # ONE buffer, nothing overlapped: every stage of encode + decode back to back on one stream (the latency a caller
            # with a single buffer sees; `value` above is the throughput of the pipelined job)
            cx, b = ctxs[0], sets[0]
            cx.use_stream(main)

            def one_buffer():
                cx.histogram(b.d_in, out=hists[0])
                cx.build_code(hists[0], codes[0])
                cx.encode_plan(b.d_in, codes[0], total=t_total[0])
                cx.decode_prepare(codes[0])
                cx.encode_emit(b.d_in, codes[0], b.out, flags=local_flags, index=b.index, end=b.end)
                cx.decode(b.out, b.bound, codes[0], b.index, d_out=b.dec, nbytes=t_nbytes)

            one_buffer()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record(main)
            for _ in range(reps):
                one_buffer()
            e1.record(main)
            torch.cuda.synchronize()
            cx.sync()
            res["single_buffer_ms"] = round(e0.elapsed_time(e1) / reps, 4)
            res["single_buffer_GBps"] = round(n / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9, 3)
            # the decode path a stream WITHOUT side-car takes (a .crs2 the reference wrote; what Decompressor<...>::decompress()
            # of the C++ host layer runs): K6 rebuilds the side-car on the GPU, then K7.  Host wall time: K6 looks at a
            # convergence word from the host a few times.
            # (THIS buffer's stream length: comp_bytes is another set's, a few hundred bytes more or less -- stale bytes behind
            #  the end mark are legal input, but K6 then iterates over a tail nobody will read: what round 3's and this round's
            #  first "4 GiB Zipf: 45..59 ms" were)
            own_bytes = int(b.end[1].item())
            cx.decode(b.out, own_bytes, codes[0], None, d_out=b.dec, cap=n, nbytes=t_nbytes)
            torch.cuda.synchronize()
            reps = 5
            tf0 = time.perf_counter()
            for _ in range(reps):
                cx.decode(b.out, own_bytes, codes[0], None, d_out=b.dec, cap=n, nbytes=t_nbytes)
            torch.cuda.synchronize()
            tf = (time.perf_counter() - tf0) / reps
            cx.sync()
            ok_f = int(t_nbytes.item()) == n and (args.no_verify or bool((b.dec[:n] == b.d_in).all().item()))
            res["decode_foreign"] = {"GBps": round(n / tf / 1e9, 3), "ms": round(tf * 1e3, 4), "round_trip_ok": bool(ok_f),
                                     "vs_indexed_decode": round(tf * 1e3 / stage_alone["decode"], 2) if stage_alone["decode"] else None,
                                     "what": "ghf_decode(index = NULL): K6 side-car reconstruction + K7, host wall time per stream"}
        if world > 1 and args.backend == "nccl":
            try:
                res["config"]["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                pass
        if world == 1 and not args.no_configs:
            # the other single-GPU BASELINE configs at full size: 4 GiB uniform (north star: encode read side), configs[2]
            # (4 GiB Zipf encode) and configs[4] (4 GiB 16-symbol decode); every number has its stage times beside it
            host_in = sets[0].d_in.cpu().numpy() if not args.no_cpu_baseline else None
            for b in sets:
                ctx.index_free(b.index)
                b.index = None
                del b.d_in, b.out, b.dec
            torch.cuda.empty_cache()
            blk = {}
            for key, kind in (("uniform_4GiB", "uniform"), ("configs[2]_zipf_4GiB", "zipf"), ("configs[4]_sym16_4GiB", "sym16")):
                try:
                    blk[key] = full_size_config(torch, ghf, synth, ctx, kind, 4096)
                except Exception as e:  # (e.g. not enough free HBM on a shared box)
                    blk[key] = {"error": repr(e)[:200]}
            res["configs"] = blk
        else:
            host_in = sets[0].d_in.cpu().numpy() if (world == 1 and not args.no_cpu_baseline) else None
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(host_in, args.kind)
        print(json.dumps(res), flush=True)
    wd.close()
    for b in sets:
        if b.index is not None:
            ctx.index_free(b.index)
    for cm in comms or []:
        ctx.comm_destroy(cm)
    for cx in reversed(ctxs):
        cx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
