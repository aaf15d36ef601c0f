#!/usr/bin/env python3
"""bench.py — whole-job queries/s of the TokenGen -> Route -> Refine hot path on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run, one rank per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from env).
One "step" = one pass of the hot path over one batch of `--batch` queries PER GPU
(weak scaling: the index is replicated, query batches shard embarrassingly):

    encode (exact fp64 Coding.H/C)  ->  route (probe + dedupe + Java-order select-B)
    -> candidate staging (device gather from a plaintext store: stand-in for the host's
       load + AES-GCM decrypt, which stay on the host in production)
    -> refine (sequential-fp64 L2 scan + stable top-k)  [-> RCCL all-gather of top-k, N > 1]

All inputs (queries, frozen index, plaintext store) are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.  Extra objects: `roofline` (refine scan, live HIP-event
timing on the context's stream), `cpu_baseline` (the C++ oracle on this box's host cores,
bounded sample), `stages_ms`, `recall_at_10`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # BASELINE.json configs[1]: SIFT-1M-shaped, 16 tables x 32 bits (m=16, lambda=2, divisions=1), B=256, batch=1024
    "sift1m_T16_b32_B256_Q1024": dict(n=1_000_000, d=128, T=16, D=1, m=16, lam=2, B=256, Q=1024, k=10),
    # BASELINE.json configs[0]: plumbing case
    "synth10k_T8_b16_B64_Q100": dict(n=10_000, d=128, T=8, D=1, m=8, lam=2, B=64, Q=100, k=10),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="sift1m_T16_b32_B256_Q1024", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="queries per GPU per step (default: workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=512, help="queries timed on the CPU oracle")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--query-batches", type=int, default=8,
                    help="distinct query batches cycled through by the steps (a repeated batch would re-read the same "
                         "candidate rows out of the 256 MiB Infinity Cache instead of HBM)")
    ap.add_argument("--merge", default="auto", choices=["auto", "inline", "overlap"],
                    help="N > 1: the RCCL all-gather of the per-rank top-k follows Refine on the same stream (inline), or runs on "
                         "a side stream overlapping the next step (overlap; costs two cross-stream events per step). auto: both "
                         "are tried for a few untimed steps before the warmup and the faster one is used by every rank")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra two-stream pass reported under 'pipelined'")
    ap.add_argument("--route-counters", action="store_true",
                    help="also produce lastCandKept / rawSeen per query (forces the full select)")
    ap.add_argument("--candidates", default="store", choices=["store", "dense"],
                    help="store: refine reads candidate rows from the resident store by id; dense: a gather kernel packs "
                         "them into [Q][B][d] first (explicit stand-in for the host's load + decrypt)")
    ap.add_argument("--streams", type=int, default=1,
                    help="contexts (HIP streams) per GPU that alternate steps; 2 overlaps the latency-bound Route of step i+1 "
                         "with the bandwidth-bound gather/refine of step i (per-stage times then include the overlap)")
    args = ap.parse_args()

    # Only the final JSON line may reach stdout: libraries (RCCL prints a version banner) write to fd 1 too.
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run the collective runs even at N = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = graft.load_package()
    wl = dict(WORKLOADS[args.workload])
    if args.batch > 0:
        wl["Q"] = args.batch
    n, d, T, D, m, lam, B, Q, k = (wl[x] for x in ("n", "d", "T", "D", "m", "lam", "B", "Q", "k"))
    TD, W = T * D, (m * lam + 63) // 64

    # ---------------- synthetic data (same on every rank; queries differ per rank) ----------------
    t0 = time.time()
    rng = np.random.default_rng(args.seed)
    X = rng.standard_normal((n, d), dtype=np.float32)
    qrng = np.random.default_rng(args.seed + 1000 + rank)
    NB = max(1, args.query_batches)
    Qall = qrng.standard_normal((NB, Q, d), dtype=np.float32)
    Qh = Qall[0]
    cfg = pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d, seed=13, refinement_limit=B)
    ctxs = []
    # a second context (own HIP stream) is set up for the extra "pipelined" pass: steps alternate between two streams,
    # so the latency-bound Route of one batch overlaps the HBM-bound Refine of the other
    want_pipe = (world == 1) and not args.no_pipelined and args.streams == 1
    for si in range(max(1, args.streams, 2 if want_pipe else 1)):
        c_ = pkg.FspannContext(cfg, local_rank)
        if si == 0:
            c_.registry_initialize(X[:1000].astype(np.float64))   # GFunctionRegistry.initialize from the first 1000 vectors
            c_.set_id_meta(n)
            c_.build_index(X)                                      # GPU coding (MFMA pre-filter + exact re-check) + partition cut
        else:                                                      # further streams import the frozen state
            c_.set_gfunctions(*ctxs[0].get_gfunctions())
            c_.set_id_meta(n)
            for td in range(TD):
                c_.set_index(td, **ctxs[0].get_index(td))
            c_.finalize()
        c_.store_set(X)                                            # plaintext store (decrypt stand-in), fp32
        ctxs.append(c_)
    ctx = ctxs[0]
    if rank == 0:
        log(f"[bench] setup {time.time() - t0:.1f}s: n={n} d={d} T*D={TD} bits={m * lam} B={B} Q/GPU={Q} k={k}")

    # ---------------- device buffers ------------------------------------------------------------------
    q_all = torch.from_numpy(Qall).to(dev)
    q_dev = q_all[0]
    from fspann_amd import dist as fdist

    class _TorchEv:          # same interface over a default torch event (system-scope fence on record)
        def __init__(self):
            self.e = torch.cuda.Event()

        def record(self, st):
            self.e.record(st)

        def wait(self, st):
            st.wait_event(self.e)

    def mkev():
        if use_dist:
            return fdist.DeviceEvent()        # device-scope release: no L2 writeback/invalidate per hand-off
        return _TorchEv()

    def mkbufs():
        return dict(codes=torch.zeros((Q, TD, W), dtype=torch.int64, device=dev), bad=torch.zeros(Q, dtype=torch.int32, device=dev),
                    sel_ids=torch.full((Q, B), -1, dtype=torch.int32, device=dev), sel_cnt=torch.zeros(Q, dtype=torch.int32, device=dev),
                    kept=torch.zeros(Q, dtype=torch.int32, device=dev), raw=torch.zeros(Q, dtype=torch.int32, device=dev),
                    cand=torch.zeros((Q, B, d), dtype=torch.float32, device=dev),
                    # results are double-buffered so the all-gather of step i (side stream) overlaps step i+1
                    # (ids and distances of one step live in ONE byte buffer: the merge is a single collective)
                    topk=[fdist.TopkBuffer(Q, k, dev) for _ in range(2)],
                    out_cnt=torch.zeros(Q, dtype=torch.int32, device=dev), scored=torch.zeros(Q, dtype=torch.int32, device=dev),
                    gathered=[fdist.GatheredTopk(world, Q, k, dev) for _ in range(2)] if use_dist else None,
                    ev_done=[mkev() for _ in range(2)], ev_gath=[mkev() for _ in range(2)], nsteps=0)

    bufs = [mkbufs() for _ in ctxs]
    out_ids, out_dist = bufs[0]["topk"][0].ids, bufs[0]["topk"][0].dist
    gathered = bufs[0]["gathered"][0] if use_dist else None
    side = torch.cuda.Stream(device=dev) if use_dist else None
    # the collective is issued straight through librccl when that works (host cost per call: us instead of ~80 us)
    rccl = fdist.DirectRccl(world, rank, dev) if use_dist else None
    if use_dist:
        flag = torch.tensor([1 if rccl.ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # all ranks or none
        if int(flag.item()) == 0:
            rccl.ok = False
    def merge(local, out, st):
        """the ONE collective of the path, issued on stream `st`"""
        if rccl.ok:
            rccl.allgather_topk(local, out, st)
        else:
            with torch.cuda.stream(st):
                fdist.allgather_topk(local, out)

    gather_path = "ncclAllGather via librccl (direct)" if (use_dist and rccl.ok) else ("torch.distributed all_gather_into_tensor" if use_dist else None)
    torch.cuda.synchronize()

    streams = [torch.cuda.ExternalStream(c_.stream, device=dev) for c_ in ctxs]
    F32 = pkg._native.F32
    step_no = [0]
    merge_mode = [args.merge if args.merge != "auto" else "inline"]   # "inline" | "overlap"; "auto" is settled before the warmup
    # QSI's adaptive retry (QSI:327-337,444-447): one more pass with 10 probes when returned < K or decrypted < 10*K.
    # With B < 10*K the second condition always holds (decrypted <= B), so every query takes both passes and the second
    # one is the answer; with B >= 10*K (and >= K finite candidates, true for the synthetic data) it never triggers.
    probe_passes = [-1, 10] if B < 10 * k else [-1]
    active = [max(1, args.streams)]          # contexts the steps alternate between
    dense = (args.candidates == "dense")

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def step(events=None, batch=None, ref_only=False):
        # events: 5 HIP events on the context's stream; ref_only = record just [3] and [4] (around the refinement scan)
        si = step_no[0] % active[0]
        qp = q_all[(step_no[0] // active[0]) % NB].data_ptr() if batch is None else q_all[batch].data_ptr()
        step_no[0] += 1
        cx, stream, b = ctxs[si], streams[si], bufs[si]
        if events is None and not dense and not args.route_counters:
            # the whole step in ONE library call (encode -> route(limit = B) -> refine from the store, stream order)
            par = b["nsteps"] & 1
            b["nsteps"] += 1
            if use_dist and b["nsteps"] > 2 and merge_mode[0] != "inline":
                b["ev_gath"][par].wait(stream)            # the all-gather that last read this result buffer has finished
            for pov in probe_passes:
                cx.search_store_dev(Q, qp, F32, pov, B, k, b["topk"][par].ids.data_ptr(), b["topk"][par].dist.data_ptr(),
                                    b["out_cnt"].data_ptr(), b["scored"].data_ptr(), b["sel_ids"].data_ptr(), b["sel_cnt"].data_ptr(),
                                    b["bad"].data_ptr())
            if use_dist and merge_mode[0] == "inline":
                # the collective follows Refine on the SAME stream: no cross-stream events (each costs this stream two
                # extra barrier packets, ~20 us per step on this runtime — more than the all-gather itself)
                merge(b["topk"][par], b["gathered"][par], stream)
                return
            if use_dist:
                b["ev_done"][par].record(stream)
                b["ev_done"][par].wait(side)
                merge(b["topk"][par], b["gathered"][par], side)
                b["ev_gath"][par].record(side)
            return
        for pov in probe_passes[:-1]:   # first pass of the adaptive retry (see probe_passes); the stages below are the last pass
            cx.search_store_dev(Q, qp, F32, pov, B, k, b["topk"][0].ids.data_ptr(), b["topk"][0].dist.data_ptr(),
                                b["out_cnt"].data_ptr(), b["scored"].data_ptr(), b["sel_ids"].data_ptr(), b["sel_cnt"].data_ptr(),
                                b["bad"].data_ptr())
        if events is not None and not ref_only:
            events[0].record(stream)
        cx.encode_dev(Q, qp, F32, b["codes"].data_ptr(), 0, b["bad"].data_ptr())
        if events is not None and not ref_only:
            events[1].record(stream)
        # lastCandKept / rawSeen are profiler counters of the reference (QSI metrics), not inputs of Refine: they are
        # only computed on request (--route-counters), which forces the full select over every probed partition
        cx.route_dev(Q, b["codes"].data_ptr(), probe_passes[-1], B, B, b["sel_ids"].data_ptr(), 0, b["sel_cnt"].data_ptr(),
                     b["kept"].data_ptr() if args.route_counters else 0, b["raw"].data_ptr() if args.route_counters else 0)
        if events is not None and not ref_only:
            events[2].record(stream)
        if dense:   # explicit stand-in for the host's load + decrypt: pack F_q rows into [Q][B][d]
            cx.store_gather_dev(Q, b["sel_ids"].data_ptr(), b["sel_cnt"].data_ptr(), B, b["cand"].data_ptr())
        if events is not None:
            events[3].record(stream)
        par = b["nsteps"] & 1
        b["nsteps"] += 1
        if use_dist and b["nsteps"] > 2 and merge_mode[0] != "inline":
            b["ev_gath"][par].wait(stream)            # the all-gather that last read this result buffer has finished
        if dense:
            cx.refine_dev(Q, qp, F32, b["cand"].data_ptr(), F32, B, b["sel_ids"].data_ptr(), b["sel_cnt"].data_ptr(), k,
                          b["topk"][par].ids.data_ptr(), b["topk"][par].dist.data_ptr(), b["out_cnt"].data_ptr(), b["scored"].data_ptr())
        else:       # candidate rows are read from the resident store by id inside the scan (each row once, no copy)
            cx.refine_store_dev(Q, qp, F32, B, b["sel_ids"].data_ptr(), b["sel_cnt"].data_ptr(), k,
                                b["topk"][par].ids.data_ptr(), b["topk"][par].dist.data_ptr(), b["out_cnt"].data_ptr(),
                                b["scored"].data_ptr())
        if events is not None:
            events[4].record(stream)
        if use_dist and merge_mode[0] == "inline":
            merge(b["topk"][par], b["gathered"][par], stream)
        elif use_dist:
            # --merge overlap: the all-gather of [Q x k] (id, dist) per rank on a side stream, so that it overlaps the
            # next step's kernels (two events per step; see --merge)
            b["ev_done"][par].record(stream)
            b["ev_done"][par].wait(side)
            merge(b["topk"][par], b["gathered"][par], side)
            b["ev_gath"][par].record(side)

    def barrier():
        for c_ in ctxs:
            c_.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    # --merge auto: which placement of the collective is faster depends on the all-gather latency of this node (ranks,
    # xGMI hops) against the fixed cost of two cross-stream events; a short untimed trial decides, rank 0's verdict holds
    merge_trial = None
    if use_dist and args.merge == "auto":
        trial = {}
        for mode in ("inline", "overlap"):
            merge_mode[0] = mode
            for b_ in bufs:
                b_["nsteps"] = 0
            for _ in range(4):
                step()
            barrier()
            t_t = time.perf_counter()
            for _ in range(12):
                step()
            barrier()
            trial[mode] = (time.perf_counter() - t_t) / 12
        tt = torch.tensor([trial["inline"], trial["overlap"]], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)                     # slowest rank per mode
        merge_mode[0] = "inline" if float(tt[0]) <= float(tt[1]) else "overlap"
        merge_trial = "inline %.1f us/step, overlap %.1f us/step" % (float(tt[0]) * 1e6, float(tt[1]) * 1e6)
        for b_ in bufs:
            b_["nsteps"] = 0
        barrier()

    for _ in range(args.warmup):
        step()
    barrier()

    evs = [[ev() for _ in range(5)] for _ in range(args.steps)]
    barrier()
    t_start = time.perf_counter()
    # timed region: every TIMED_EVERY-th refinement-scan dispatch carries its own start/stop HIP events (kernel-attached, on the
    # context's stream) -> roofline.  (An attached pair costs a few us of stream time, so not every dispatch gets one.)
    TIMED_EVERY = max(2, args.steps // 8)     # about eight timed dispatches whatever --steps is
    for c_ in ctxs[:active[0]]:
        c_.refine_timing_begin(args.steps, TIMED_EVERY)
    for i in range(args.steps):
        step()
    for c_ in ctxs:
        c_.sync()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    rt = [c_.refine_timing_end() for c_ in ctxs[:active[0]]]
    ref_ms_timed = sum(t for _, t in rt) / max(1, sum(n for n, _ in rt))          # kernel-attached events

    # ---- stage breakdown: a separate, untimed pass with an event after every stage -----------------------------
    nprof = min(args.steps, 20)
    barrier()
    for i in range(nprof):
        step(evs[i])
    barrier()

    # ---- extra pass: the same steps alternating between two streams (reported, never `value`) ----
    pipelined = None
    if want_pipe:
        active[0] = 2
        step_no[0] = 0
        for b_ in bufs:
            b_["nsteps"] = 0
        for _ in range(max(2, args.warmup)):
            step()
        barrier()
        t_p = time.perf_counter()
        for i in range(args.steps):
            step()
        for c_ in ctxs:
            c_.sync()
        torch.cuda.synchronize()
        el_p = time.perf_counter() - t_p
        pipelined = dict(streams=2, value=round(Q * args.steps / el_p, 1), unit="queries/s", ms_per_step=round(el_p * 1000.0 / args.steps, 4),
                         note="same steps alternating between two contexts/HIP streams on this GPU: Route of one batch overlaps "
                              "Refine of the other; kernel durations are no longer solo, so the roofline above is not taken here")
        active[0] = max(1, args.streams)

    # one more (untimed) step of batch 0 on context 0: its results are what recall and the CPU baseline are checked on
    step_no[0] = 0
    bufs[0]["nsteps"] = 0
    barrier()
    step(batch=0)
    barrier()

    stage_ms = np.array([[evs[i][j].elapsed_time(evs[i][j + 1]) for j in range(4)] for i in range(nprof)])
    st_mean = stage_ms.mean(axis=0)
    ms_per_step = elapsed * 1000.0 / args.steps
    qps = world * Q * args.steps / elapsed

    # ---------------- roofline of the refinement scan (north_star's HBM-bound kernel) -----------------
    # algorithmic bytes per launch: Q * (B*d*4 + d*4 + k*8)
    ref_bytes = Q * (B * d * 4 + d * 4 + k * 8)
    ref_ms = ref_ms_timed                  # the refinement scan's launches inside the timed region
    achieved = ref_bytes / (ref_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "refine_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            tj = tj.get(args.candidates, {})
            if tj.get("workload") == args.workload and tj.get("Q") == Q:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    kname = "refine_scan_kernel<float,float,32,true,%s>" % ("false" if dense else "true")
    roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 1),
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    algorithmic_bytes_per_launch=ref_bytes, avg_launch_ms=round(ref_ms, 5), launches=sum(n for n, _ in rt),
                    timing="HIP start/stop events attached to every %d-th refine_scan_kernel dispatch of the timed region "
                           "(hipExtLaunchKernel, on the context's stream)" % TIMED_EVERY,
                    bracket_ms=round(float(st_mean[3]), 5))

    # Route (probe + select) is the longest stage but is bound by dependent L2 rounds and LDS atomics, not by HBM or
    # MFMA; its algorithmic bytes (per (t,d) search + rep/id-range fetch + P*S ids) are reported for scale.
    P_, S_ = 5, 64
    nparts = (n + S_ - 1) // S_
    levels = max(1, int(np.ceil(np.log(max(nparts, 2)) / np.log(16))))
    route_bytes = Q * (TD * (levels * 16 * 16 + (2 * P_ - 1) * (8 * W + 8) + P_ * S_ * 4) + B * 4)
    route_ms = float(st_mean[1])
    rinfo = ctx.last_route_info()
    route_info = dict(kernels="route_probe_kernel + " + ("route_select_lazy_kernel<256> (bounded select; %d of %d queries handed to the full select)"
                                                         % (rinfo["overflowed"], Q) if rinfo["lazy"] else "route_select_kernel<true,512>"), bound="L2 latency + LDS atomics (integer)",
                      avg_ms=round(route_ms, 5), algorithmic_bytes_per_launch=int(route_bytes),
                      achieved_GBs=round(route_bytes / (route_ms * 1e-3) / 1e9, 1))

    # ---------------- recall@10 vs exact kNN of the synthetic set (reported, never assumed) ------------
    recall = None
    if rank == 0:
        with torch.no_grad():
            Xd = torch.from_numpy(X).to(dev)
            xx = (Xd * Xd).sum(1)
            gt = []
            for s in range(0, Q, 256):
                qq = q_dev[s:s + 256]
                dd = xx[None, :] - 2.0 * (qq @ Xd.T)
                gt.append(torch.topk(dd, k, dim=1, largest=False).indices)
            gt = torch.cat(gt).cpu().numpy()
            got = out_ids.cpu().numpy()
            recall = float(np.mean([len(set(gt[i]) & set(got[i])) / k for i in range(Q)]))
            del Xd

    # ---------------- CPU baseline: the oracle on this box's host cores (rank 0, N = 1 only) -----------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        O = graft.load_oracle()
        o = O.Oracle(T, D, m, lam, d, refinement_limit=B)
        a, r_, w_ = ctx.get_gfunctions()
        o.set_gfunctions(a, r_, w_)
        o.set_id_meta(n)
        X64 = X.astype(np.float64)
        o.set_store(X64)
        t_ob = time.perf_counter()
        o.build_index(X64)                     # the checker cuts its OWN partitions (nothing imported from the GPU build)
        t_ob = time.perf_counter() - t_ob
        del X64
        index_same = all(np.array_equal(ctx.get_index(td)[k_], v_) for td in range(TD) for k_, v_ in o.get_index(td).items())
        ns = min(args.cpu_sample, Q)
        qs = Qh[:ns].astype(np.float64)
        t1 = time.perf_counter()
        cds = o.encode(qs[:8])  # warm
        ref = o.search(qs[:8], k, codes=cds, threads=1)
        t1 = time.perf_counter()
        reps, done = 0, 0
        while True:  # bounded: at least one pass, at most ~10 s
            cds = o.encode(qs)
            ref = o.search(qs, k, codes=cds, threads=1)
            reps += 1
            done += ns
            if time.perf_counter() - t1 > 10.0 or reps >= 50:
                break
        cpu_s = time.perf_counter() - t1
        same = bool(np.array_equal(ref["ids"], out_ids.cpu().numpy()[:ns]) and
                    np.array_equal(ref["dist"], out_dist.cpu().numpy()[:ns]))
        if o.unmodelled:
            raise SystemExit("bench: a HashMap bin treeified in the oracle at this workload: the checker has no pinned order")
        if not (same and index_same):
            raise SystemExit(f"bench: GPU results differ from the CPU oracle (index_same={index_same}, results_same={same})")
        cpu = dict(value=round(done / cpu_s, 1), unit="queries/s", cores=1, kind="port",
                   sample=f"{ns} queries x {reps} passes of the same batch (encode + Route + Refine on plaintext, "
                          f"no AES/RocksDB), C++ oracle single thread; host has {os.cpu_count()} logical cores",
                   matches_gpu=same, index_matches_gpu=bool(index_same), oracle_unmodelled=bool(o.unmodelled),
                   oracle_index_build_s=round(t_ob, 1))
        # the same port over the host cores this GPU's share allows (queries are independent: threads over queries)
        nthr = max(1, min(16, os.cpu_count() or 1))
        if nthr > 1:
            t2 = time.perf_counter()
            reps2, done2 = 0, 0
            while True:
                cds = o.encode(qs)
                o.search(qs, k, codes=cds, threads=nthr)
                reps2 += 1
                done2 += ns
                if time.perf_counter() - t2 > 5.0 or reps2 >= 200:
                    break
            cpu["multi_thread"] = dict(value=round(done2 / (time.perf_counter() - t2), 1), unit="queries/s", cores=nthr,
                                       note="Route + Refine threaded over queries, encode single-threaded")

    if rank == 0:
        out = {
            "metric": "queries/sec @ recall@10, SIFT-1M d=128 B=256",
            "value": round(qps, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic N(0,1) fp32 vectors (SIFT-1M shape), exact-kNN ground truth of the synthetic set",
            "config": {"workload": args.workload, "n": n, "dim": d, "tables": T, "divisions": D, "m": m, "lambda": lam,
                       "code_bits": m * lam, "probes": 5, "B": B, "k": k, "queries_per_gpu_per_step": Q, "distinct_query_batches": NB, "route_counters": bool(args.route_counters), "passes_per_step": len(probe_passes),
                       "parallelism": f"query-sharded x{world}, index replicated", "merge": (gather_path + (", same stream" if merge_mode[0] == "inline" else ", side stream") + (" (auto: %s)" % merge_trial if merge_trial else "")) if use_dist else None, "streams_per_gpu": active[0],
                       "candidates": "rows read from the HBM-resident plaintext store by id inside the refine scan" if not dense
                       else "rows packed into [Q][B][d] by a gather kernel (host decrypt stand-in), then scanned"},
            "recall_at_10": recall,
            "stages_ms": {"encode": round(float(st_mean[0]), 5), "route_select": round(float(st_mean[1]), 5),
                          "stage_candidates": round(float(st_mean[2]), 5), "refine_topk": round(float(st_mean[3]), 5)},
            "roofline": roofline,
            "route_stage": route_info,
            "pipelined": pipelined,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist and rank == 0:
        # merged result = every rank's top-k in rank order; rank 0's own slice must be intact
        g_ids, g_dist = gathered.split()
        assert torch.equal(g_ids[:Q], out_ids) and torch.equal(g_dist[:Q], out_dist)
    for c_ in ctxs:
        c_.close()
    if use_dist:
        rccl.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
