#!/usr/bin/env python3
"""bench.py — whole-job queries/s of FSPANN's TokenGen -> Route -> Refine hot path on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torch.distributed.run, one rank per
GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the env).  One "step" = one pass of the hot path over one batch of
`--batch` queries PER GPU (weak scaling: the index is replicated, query batches shard embarrassingly):

    encode  (Coding.H/C, sequential fp64)                                   idx/Coding.java:250-301
 -> route   (probe + HashMap-ordered candidate list, first B = stage A.5)   PIS:592-715, QSI:169-214
 -> refine  (sequential-fp64 L2 scan over the candidate rows + stable top-k) QSI:238-316
 [-> one RCCL all-gather of the per-rank top-k, N > 1]

`value` is the KERNEL PATH of SURVEY §8(d): the [Q x B x d] blocks of decrypted candidate rows — what the host's
loadPointIfActive + AES-GCM decrypt loop hands over in production — are already resident in HBM when the timed region
starts (`--candidates dense`; packed once, before the timed region, for each of the `--query-batches` distinct batches,
so consecutive steps read DIFFERENT blocks: 32 x 134 MB = 4.3 GB cycled, far beyond the 256 MiB Infinity Cache).
Beside it (extra objects, never `value`): the trusted-HBM variant where Refine reads plaintext rows of a resident store
by id (`variants.store`), the gather-inclusive variant, the single-stream and single-kernel pipelines, the refine scan on an
8 M-row store and at config #4's shape (`roofline.hbm_proof`, `roofline.cfg4_shape`), recall@10 / distance ratio vs
exact kNN, and the CPU oracle on this box's host cores (`cpu_baseline`).  Rank 0 prints ONE JSON line.
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # BASELINE.json configs[1]: SIFT-1M-shaped, 16 tables x 32 bits (m=16, lambda=2, divisions=1), B=256, batch=1024
    "sift1m_T16_b32_B256_Q1024": dict(n=1_000_000, d=128, T=16, D=1, m=16, lam=2, B=256, Q=1024, k=10),
    # BASELINE.json configs[0]: plumbing case
    "synth10k_T8_b16_B64_Q100": dict(n=10_000, d=128, T=8, D=1, m=8, lam=2, B=64, Q=100, k=10),
    # BASELINE.json configs[2] / [3]: ONE GPU's shard of the 8-GPU configurations (queries are sharded, the index is replicated:
    # 4 096 / 8 and 8 192 / 8 queries per GPU and step) — parity-test cases first (tests/test_gpu_fullsize.py), measured here on request
    "gist1m_T16_b32_B512_Q512": dict(n=1_000_000, d=960, T=16, D=1, m=16, lam=2, B=512, Q=512, k=10, nb=8),
    "synth10m_T32_b64_B1024_Q1024": dict(n=10_000_000, d=768, T=32, D=1, m=32, lam=2, B=1024, Q=1024, k=10, nb=2),
    # The profiles the reference SHIPS and publishes its numbers at (config/src/main/resources/config_sift1m.json:44-128,
    # logs/New Results:27-57): k = 100 is eval.kVariants' maximum = the token's topK.  HARD_CAP =
    # max(maxGlobalCandidates, refinementLimit) is below T*D*P*64 for both, so the cap can cut the traversal; at P10_HIGH bestScore
    # also resizes 32 768 -> 65 536.  Full select + chunked scan + merge; dense blocks of 4.2 / 11.5 GB per batch.
    "sift1m_P4_FAST": dict(n=1_000_000, d=128, T=5, D=8, m=20, lam=2, B=8000, Q=1024, k=100, probes=4, hard_cap=10000, nb=6, shipped=True),
    "sift1m_P10_HIGH": dict(n=1_000_000, d=128, T=7, D=8, m=26, lam=2, B=22000, Q=1024, k=100, probes=10, hard_cap=28000, nb=3, shipped=True),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def read_fvecs(path, limit=None):
    """.fvecs: per record int32 dim + dim float32 (little endian), loader/FvecsLoader.java:21-35 (values are then widened
    float -> double, :29: the data are exactly fp32-representable, which is what the scan's bit-exactness relies on)."""
    a = np.fromfile(path, dtype=np.int32)
    if a.size == 0:
        raise ValueError(f"{path}: empty")
    d = int(a[0])
    if d <= 0 or a.size % (d + 1) != 0:
        raise ValueError(f"{path}: not an fvecs file (dim {d}, {a.size} words)")
    a = a.reshape(-1, d + 1)
    if not (a[:, 0] == d).all():
        raise ValueError(f"{path}: records of different dimension")
    out = a[:, 1:].view(np.float32)
    return np.ascontiguousarray(out[:limit] if limit else out)


def read_ivecs(path, limit=None):
    """.ivecs ground truth: per record int32 k + k int32 ids (loader/GroundtruthManager.java:69-146)."""
    a = np.fromfile(path, dtype=np.int32)
    k = int(a[0])
    a = a.reshape(-1, k + 1)
    return np.ascontiguousarray(a[:limit, 1:] if limit else a[:, 1:])


def make_data(kind, n, d, nb, q, seed, rank):
    """Base vectors (same on every rank) and nb x q queries (per rank).  gaussian: iid N(0,1) (north_star's throughput data;
    no LSH has signal there).  clustered: 4096 Gaussian blobs (sigma 0.15 around N(0,1) centres), queries from the same
    mixture — synthetic data on which recall means something.  <path>.fvecs: real vectors, queries from <path> with
    'base' -> 'query' in the name when present."""
    rng = np.random.default_rng(seed)
    qrng = np.random.default_rng(seed + 1000 + rank)
    if kind == "gaussian":
        X = rng.standard_normal((n, d), dtype=np.float32)
        Qs = qrng.standard_normal((nb, q, d), dtype=np.float32)
        return X, Qs, "synthetic N(0,1) fp32 vectors (SIFT-1M shape)"
    if kind == "clustered":
        nc = 4096
        C = rng.standard_normal((nc, d), dtype=np.float32)
        X = C[rng.integers(0, nc, n)] + np.float32(0.15) * rng.standard_normal((n, d), dtype=np.float32)
        Qs = C[qrng.integers(0, nc, nb * q)] + np.float32(0.15) * qrng.standard_normal((nb * q, d), dtype=np.float32)
        return X, Qs.reshape(nb, q, d), "synthetic clustered fp32 vectors (4096 Gaussian blobs, sigma 0.15)"
    if kind.startswith("siftlike"):
        # SIFT-like difficulty without the file: intrinsic dimension r (local intrinsic dimensionality estimates of SIFT1M are ~16),
        # embedded in d dimensions, plus a little full-rank noise, shifted / scaled / rounded to integers 0..255 like .fvecs SIFT data
        parts = kind.split(":")
        r = int(parts[1]) if len(parts) > 1 else 16
        noise = float(parts[2]) if len(parts) > 2 else 6.0
        U = (rng.standard_normal((r, d)) / np.sqrt(r)).astype(np.float32)
        def draw(g, cnt):
            out = np.empty((cnt, d), dtype=np.float32)
            for a in range(0, cnt, 65536):              # in pieces: the host copy of 1 M x 128 is enough, no second one
                b = min(cnt, a + 65536)
                y = g.standard_normal((b - a, r), dtype=np.float32) @ U
                y = np.float32(64.0) + np.float32(48.0) * y + np.float32(noise) * g.standard_normal((b - a, d), dtype=np.float32)
                out[a:b] = np.clip(np.rint(y), 0, 255)
            return out
        X = draw(rng, n)
        Qs = draw(qrng, nb * q)
        return X, Qs.reshape(nb, q, d), "synthetic SIFT-like fp32 vectors (integers 0..255, intrinsic dimension %d + noise %g)" % (r, noise)
    X = read_fvecs(kind, n)
    if X.shape[1] != d:
        raise SystemExit(f"{kind}: dim {X.shape[1]} != workload dim {d}")
    qpath = kind.replace("base", "query")
    if qpath != kind and os.path.exists(qpath):
        Qf = read_fvecs(qpath)
    else:
        Qf = X[qrng.integers(0, len(X), nb * q)] + np.float32(0.01) * qrng.standard_normal((nb * q, d), dtype=np.float32)
    reps = -(-nb * q // len(Qf))
    Qs = np.tile(Qf, (reps, 1))[: nb * q].reshape(nb, q, d)
    return X, np.ascontiguousarray(Qs), f"{os.path.basename(kind)} ({len(X)} x {d}), queries tiled to {nb} x {q}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="sift1m_T16_b32_B256_Q1024", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="queries per GPU per step (default: workload's)")
    ap.add_argument("--data", default="gaussian", help="gaussian | clustered | path to a .fvecs base file")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=512, help="queries timed on the CPU oracle")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra passes (store / gather variants, pipelined, HBM proof)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--k", type=int, default=0, help="top-k (default: the workload's)")
    ap.add_argument("--query-batches", type=int, default=0,
                    help="distinct query batches (and dense candidate blocks) cycled through by the steps: 32 x 134 MB = 4.3 GB, "
                         "so a step never finds its rows in the 256 MiB Infinity Cache")
    ap.add_argument("--candidates", default="dense", choices=["dense", "store", "gather"],
                    help="dense (value): Refine scans the [Q][B][d] block of decrypted rows resident in HBM, packed before the timed "
                         "region; store: Refine reads rows of a resident plaintext store by id (trusted-HBM variant, one library "
                         "call per step); gather: a kernel packs the block inside the step (device stand-in for load + decrypt)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch queries per GPU per step; strong: ONE batch of the workload's size cut into contiguous shards")
    ap.add_argument("--merge", default="inline", choices=["inline", "off"], help="N > 1: all-gather of the top-k behind Refine, same stream")
    ap.add_argument("--merge-every", type=int, default=4,
                    help="N > 1: a context gathers the top-k of this many of its batches with ONE collective (bigger, fewer collectives; "
                         "1 = one per batch); every batch of the timed region is gathered inside it")
    ap.add_argument("--route-counters", action="store_true", help="also produce lastCandKept / rawSeen (forces the full select)")
    ap.add_argument("--host-threads", type=int, default=16, help="host threads of the end-to-end pipeline's AES-GCM pool (0: every core this process may use)")
    ap.add_argument("--prewarm", type=int, default=400, help="untimed steps of the same pipeline before the --warmup steps (device clocks and caches)")
    ap.add_argument("--contexts", type=int, default=3, help="contexts (each with its own HIP stream) per GPU for --pipeline concurrent")
    ap.add_argument("--pipeline", default="front", choices=["concurrent", "front", "serial", "tick"],
                    help="concurrent: --contexts independent contexts per GPU take the batches in turn, each running encode -> "
                         "Route -> Refine of ITS batch as three kernels on its own HIP stream; the hardware queues overlap the "
                         "latency-bound Route of one batch with the bandwidth-bound Refine of another.  serial: one context, one stream "
                         "(latency of a single batch, nothing overlaps).  tick: one stream, three batches in flight, encode(t+2) + "
                         "Route(t+1) + Refine(t) as ONE kernel (fspann_tick_dev).  front (value): as concurrent, but a context runs the encode "
                         "of its NEXT batch and the Route of this one as ONE launch (fspann_tick_dev without a Refine part), then the scan")
    ap.add_argument("--no-shipped", action="store_true", help="skip the reference's shipped profiles (SIFT_P4_FAST / SIFT_P10_HIGH), run as child "
                                                              "processes after the headline workload and reported under `extra`")
    ap.add_argument("--shipped-data", default="siftlike:16:6", help="data of the shipped-profile child runs (make_data kinds)")
    ap.add_argument("--shipped-steps", type=int, default=40, help="timed steps of each shipped-profile child run")
    ap.add_argument("--solo-tail", type=int, default=16, help="drained solo dispatches of the refinement scan timed AFTER the timed region "
                                                              "(the roofline's readings do not depend on --steps)")
    ap.add_argument("--solo-in-region", action="store_true",
                    help="also take drained solo readings INSIDE the timed region (every ~40th step drains all contexts and runs alone: rounds 2-3; "
                         "costs the region about two steps per reading).  Default: the solo readings come from the untimed tail (--solo-tail), the "
                         "timed region holds exactly --steps pipeline steps and only the overlapped readings")
    ap.add_argument("--launch-check", action="store_true", help="(tests) only prove the N-rank launch: every rank joins a gloo group, rank 0 prints one JSON line")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks ourselves — a CHILD torch.distributed.run,
    # spawned before this process touches the GPU (no exec: replacing a process that initialised the GPU takes the box down), its
    # stdout (rank 0's JSON line) and exit code relayed.  Never a silent one-GPU run.
    if args.gpus > 1 and "RANK" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("[bench] --gpus %d without a rank environment: launching %s" % (args.gpus, " ".join(cmd)))
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        proc = subprocess.run(cmd, env=env)
        raise SystemExit(proc.returncode)
    if args.launch_check:
        import torch.distributed as dist
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if world != args.gpus:
            raise SystemExit(f"bench: WORLD_SIZE={world} but --gpus {args.gpus}")
        if world > 1 or "RANK" in os.environ:
            import torch
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.tensor([rank + 1], dtype=torch.int64)
            dist.all_reduce(t)
            ok = int(t.item()) == world * (world + 1) // 2
            dist.destroy_process_group()
        else:
            ok = True
        if os.environ.get("FSPANN_BENCH_LAUNCH_FAIL") == str(rank):
            raise SystemExit(3)
        if rank == 0:
            print(json.dumps({"launch_check": bool(ok), "n_gpus": world}), flush=True)
        raise SystemExit(0 if ok else 1)

    # Only the final JSON line may reach stdout: libraries (RCCL prints a version banner) write to fd 1 too.
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench: WORLD_SIZE={world} but --gpus {args.gpus}: refusing to measure a different job than the one asked for")
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
    from fspann_amd import dist as fdist
    F32 = pkg._native.F32
    wl = dict(WORKLOADS[args.workload])
    if args.batch > 0:
        wl["Q"] = args.batch
    if args.k > 0:
        wl["k"] = args.k
    n, d, T, D, m, lam, B, Qw, k = (wl[x] for x in ("n", "d", "T", "D", "m", "lam", "B", "Q", "k"))
    TD, W = T * D, (m * lam + 63) // 64
    shipped = bool(wl.get("shipped"))
    P_cfg = int(wl.get("probes", 0))                 # runtime.probeOverride of the profile (0: the default 5)
    P_eff = P_cfg if P_cfg > 0 else 5
    HARD_CAP = max(int(wl.get("hard_cap", 20000)), B)
    if args.query_batches <= 0:
        args.query_batches = int(wl.get("nb", 32))
    if B > 256 and args.pipeline == "front":
        args.pipeline = "concurrent"                 # the front launch is the bounded select's (limit <= 512): three kernels per step here
    if args.scaling == "strong":       # one batch of Qw queries for the whole job, contiguous shards (last one padded)
        Q = -(-Qw // world)
        q_lo, q_hi = fdist.shard_bounds(Qw, world, rank)
    else:
        Q = Qw
        q_lo, q_hi = 0, Q
    q_live = q_hi - q_lo
    extras = (world == 1) and not args.no_extras and not shipped     # the extra passes are sized for the headline workload
    dense = args.candidates == "dense"

    # ---------------- data + frozen index ----------------------------------------------------------------------------
    t0 = time.time()
    NB = max(1, args.query_batches)
    if args.scaling == "strong":
        X, Qglob, data_desc = make_data(args.data, n, d, NB, Qw, args.seed, 0)      # the same global batches on every rank
        Qall = np.zeros((NB, Q, d), np.float32)
        Qall[:, :q_live] = Qglob[:, q_lo:q_hi]
    else:
        X, Qall, data_desc = make_data(args.data, n, d, NB, Q, args.seed, rank)
    n = len(X)
    cfg = pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d, seed=13, refinement_limit=B,
                                 max_global_candidates=int(wl.get("hard_cap", 20000)), probe_override=P_cfg if P_cfg > 0 else -1)
    ctx = pkg.FspannContext(cfg, local_rank)
    ctx.registry_initialize(X[:1000].astype(np.float64))   # GFunctionRegistry.initialize from the first 1000 vectors
    ctx.set_id_meta(n)
    t_b = time.time()
    ctx.build_index(X)                                      # GPU coding (MFMA pre-filter + exact re-check) + partition cut
    ctx.sync()
    setup_build_s = time.time() - t_b
    ctx.store_set(X)                                        # plaintext rows: source of the dense blocks / the store variant
    ctxs = [ctx]
    nctx = max(1, args.contexts) if args.pipeline in ("concurrent", "front") else 1
    for _ in range(nctx - 1):                               # further contexts (own HIP streams) reading the SAME frozen index in HBM
        ctxs.append(ctx.clone())
    if rank == 0:
        log(f"[bench] setup {time.time() - t0:.1f}s: n={n} d={d} T*D={TD} bits={m * lam} B={B} Q/GPU={Q} k={k} data={args.data}")

    # QSI's adaptive retry (QSI:327-337,444-447): one more pass with 10 probes when returned < K or decrypted < 10*K.
    # With B < 10*K the second condition always holds (decrypted <= B), so every query takes both passes and the second
    # one is the answer; with B >= 10*K (and >= K finite candidates, true for the synthetic data) it never triggers.
    probe_passes = [-1, 10] if B < 10 * k else [-1]

    # ---------------- device buffers -------------------------------------------------------------------------------------
    q_all = torch.from_numpy(Qall).to(dev)

    MG = max(1, args.merge_every) if use_dist else 1      # batches of one context per collective (packed buffers hold MG batches)

    def mkbufs():
        return dict(codes=torch.zeros((Q, TD, W), dtype=torch.int64, device=dev), bad=torch.zeros(Q, dtype=torch.int32, device=dev),
                    sel_ids=torch.full((Q, B), -1, dtype=torch.int32, device=dev), sel_cnt=torch.zeros(Q, dtype=torch.int32, device=dev),
                    kept=torch.zeros(Q, dtype=torch.int32, device=dev), raw=torch.zeros(Q, dtype=torch.int32, device=dev),
                    cand=torch.zeros((Q, B, d), dtype=torch.float32, device=dev) if args.candidates == "gather" or extras else None,
                    # packed (ids | distances) results, double-buffered; the merge is a single collective on them
                    topk=[fdist.TopkBuffer(Q * MG, k, dev) for _ in range(2)],
                    out_cnt=torch.zeros(Q, dtype=torch.int32, device=dev), scored=torch.zeros(Q, dtype=torch.int32, device=dev),
                    gathered=[fdist.GatheredTopk(world, Q * MG, k, dev) for _ in range(2)] if use_dist else None, nsteps=0,
                    # tick pipeline: three batches in flight, each with its codes, F_q and the hand-over buffer of its Route
                    slot=[dict(codes=torch.zeros((Q, TD, W), dtype=torch.int64, device=dev), bad=torch.zeros(Q, dtype=torch.int32, device=dev),
                               sel_ids=torch.full((Q, B), -1, dtype=torch.int32, device=dev), sel_cnt=torch.zeros(Q, dtype=torch.int32, device=dev),
                               hov=torch.zeros(max(1, ctx.route_handover_bytes(Q, probe_passes[-1])), dtype=torch.uint8, device=dev))
                          for _ in range(3)], tick_no=0,
                    # front pipeline: the codes of this context's current and next batch
                    fcodes=[torch.zeros((Q, TD, W), dtype=torch.int64, device=dev) for _ in range(2)], front_no=0, last_front=None,
                    # ... and the hand-over buffer its Route writes and its scan consumes (a query the bounded select cannot hold is
                    # finished by the scan's own workgroups: no hand-back launch)
                    fhov=torch.zeros(max(1, ctx.route_handover_bytes(Q, probe_passes[-1])), dtype=torch.uint8, device=dev))

    bufs = [mkbufs() for _ in ctxs]
    streams = [torch.cuda.ExternalStream(c_.stream, device=dev) for c_ in ctxs]

    # dense candidate blocks: F_q of every distinct batch, packed ONCE before the timed region by the gather kernel — the
    # device stand-in for the host's loadPointIfActive + decryptFromPoint (PIS:717-724, AesGcmCryptoService.java:126-166)
    cand_all = None
    treeified_setup = 0
    if dense or extras:
        try:
            cand_all = torch.empty((NB, Q, B, d), dtype=torch.float32, device=dev)
        except RuntimeError:
            raise SystemExit(f"cannot hold {NB} dense candidate blocks ({NB * Q * B * d * 4 / 1e9:.1f} GB): lower --query-batches")
        b0 = bufs[0]
        for bi in range(NB):
            qp = q_all[bi].data_ptr()
            cand_all[bi].zero_()
            torch.cuda.synchronize()
            ctx.encode_dev(Q, qp, F32, b0["codes"].data_ptr(), 0, b0["bad"].data_ptr())
            ctx.route_dev(Q, b0["codes"].data_ptr(), probe_passes[-1], B, B, b0["sel_ids"].data_ptr(), 0, b0["sel_cnt"].data_ptr(), 0, 0)
            # (rare) queries whose bestScore map treeifies a bin are finished by the library's host model before their rows are packed
            treeified_setup += ctx.route_resolve_dev(Q, b0["codes"].data_ptr(), probe_passes[-1], B, B, b0["sel_ids"].data_ptr(), 0, b0["sel_cnt"].data_ptr())
            ctx.store_gather_dev(Q, b0["sel_ids"].data_ptr(), b0["sel_cnt"].data_ptr(), B, cand_all[bi].data_ptr())
        ctx.sync()
        if ctx.unmodelled_queries() != 0:
            raise SystemExit("bench: a query stayed unmodelled at this workload (equal hashCodes of non-decimal ids in a tree bin)")

    comms = None
    gather_path = None
    if use_dist and args.merge != "off":
        # fspann_comm_* of the C library: one communicator per context (its all-gathers run on that context's stream),
        # bootstrapped over the torch group; all ranks fall back to torch.distributed together if any of them cannot
        comms = [fdist.LibComm(c_, world, rank, dev) for c_ in ctxs]
        lib_ok = all(cm.ok for cm in comms)
        gather_path = (f"fspann_allgather_topk_dev (ncclAllGather via {os.path.basename(comms[0].library)}) on each context's stream, "
                       f"one collective per {MG} batches of a context"
                       if lib_ok else "torch.distributed all_gather_into_tensor")

    def out_slot(b):
        """packed top-k destination of the context's next batch: (parity, ids pointer, distances pointer, last batch of its group)"""
        n = b["nsteps"]
        b["nsteps"] += 1
        par, slot = (n // MG) & 1, n % MG
        tk = b["topk"][par]
        return par, tk.ids.data_ptr() + slot * Q * k * 4, tk.dist.data_ptr() + slot * Q * k * 8, slot == MG - 1

    def merge(b, par, stream, si=0, last=True):
        """the ONE collective of the path, behind Refine on the same stream: the top-k of the context's last MG batches at once"""
        if comms is None or not last:
            return
        if comms[si].ok:
            comms[si].allgather_topk(b["topk"][par], b["gathered"][par])
        else:
            with torch.cuda.stream(stream):
                fdist.allgather_topk(b["topk"][par], b["gathered"][par])

    torch.cuda.synchronize()
    step_no = [0]
    use_front = [False]   # set by timed() for the region it times (the untimed passes use the plain three-launch step)
    active = [1]          # contexts the steps alternate between
    overlapped = [None]   # (dispatches, mean ms) of sampled refinement scans that ran next to other contexts' kernels
    host_issue_s = [None] # seconds the host needed to ISSUE the launches of the last timed() region (before any synchronisation)

    def step(mode, events=None, batch=None, force_ctx=None):
        """One pass of the hot path over one batch.  mode: dense | store | gather.  events: 5 torch events recorded around
        the stages (untimed breakdown passes only)."""
        si = step_no[0] % active[0] if force_ctx is None else force_ctx
        bi = step_no[0] % NB if batch is None else batch      # consecutive steps (whatever their context) take DIFFERENT batches
        step_no[0] += 1
        qp = q_all[bi].data_ptr()
        cx, stream, b = ctxs[si], streams[si], bufs[si]
        par, ids_p, dist_p, last = out_slot(b)
        if use_front[0] and events is None and batch is None and mode == "dense":
            # ONE launch for encode(next batch of this context) + Route(this batch), then the scan of this batch's block
            # context si's j-th step takes batch j * contexts + si: the contexts work on DIFFERENT batches at any moment
            j = b["front_no"]
            b["front_no"] += 1
            fb, nb_ = (j * active[0] + si) % NB, ((j + 1) * active[0] + si) % NB
            cx.tick_dev(encode=dict(nq=Q, q=q_all[nb_].data_ptr(), codes=b["fcodes"][(j + 1) & 1].data_ptr(), bad=b["bad"].data_ptr()),
                        route=dict(nq=Q, codes=b["fcodes"][j & 1].data_ptr(), limit=B, probe_override=probe_passes[-1], ids=b["sel_ids"].data_ptr(),
                                   count=b["sel_cnt"].data_ptr(), handover=b["fhov"].data_ptr()),
                        refine=None)
            cx.tick_dev(refine=dict(nq=Q, q=q_all[fb].data_ptr(), B=B, ids=b["sel_ids"].data_ptr(), count=b["sel_cnt"].data_ptr(), k=k,
                                    cand=cand_all[fb].data_ptr(), codes=b["fcodes"][j & 1].data_ptr(), handover=b["fhov"].data_ptr(),
                                    probe_override=probe_passes[-1], out_ids=ids_p, out_dist=dist_p, out_count=b["out_cnt"].data_ptr(),
                                    scored=b["scored"].data_ptr()))
            b["last_front"] = (fb, par, (b["nsteps"] - 1) % MG)
            merge(b, par, stream, si, last)
            return
        if mode == "store" and events is None and not args.route_counters:
            # the whole step in ONE library call (encode -> bounded select -> refine from the store, stream order)
            for pov in probe_passes:
                cx.search_store_dev(Q, qp, F32, pov, B, k, ids_p, dist_p, b["out_cnt"].data_ptr(),
                                    b["scored"].data_ptr(), b["sel_ids"].data_ptr(), b["sel_cnt"].data_ptr(), b["bad"].data_ptr())
            merge(b, par, stream, si, last)
            return
        for pov in probe_passes[:-1]:   # first pass of the adaptive retry (see probe_passes); the stages below are the last pass
            cx.search_store_dev(Q, qp, F32, pov, B, k, ids_p, dist_p, b["out_cnt"].data_ptr(),
                                b["scored"].data_ptr(), b["sel_ids"].data_ptr(), b["sel_cnt"].data_ptr(), b["bad"].data_ptr())
        if events is not None:
            events[0].record(stream)
        cx.encode_dev(Q, qp, F32, b["codes"].data_ptr(), 0, b["bad"].data_ptr())
        if events is not None:
            events[1].record(stream)
        # lastCandKept / rawSeen are profiler counters of the reference (QSI metrics), not inputs of Refine: computed only
        # on request (--route-counters), which forces the full select over every probed partition
        cx.route_dev(Q, b["codes"].data_ptr(), probe_passes[-1], B, B, b["sel_ids"].data_ptr(), 0, b["sel_cnt"].data_ptr(),
                     b["kept"].data_ptr() if args.route_counters else 0, b["raw"].data_ptr() if args.route_counters else 0)
        if events is not None:
            events[2].record(stream)
        if mode == "gather":
            cx.store_gather_dev(Q, b["sel_ids"].data_ptr(), b["sel_cnt"].data_ptr(), B, b["cand"].data_ptr())
        if events is not None:
            events[3].record(stream)
        if mode == "store":
            cx.refine_store_dev(Q, qp, F32, B, b["sel_ids"].data_ptr(), b["sel_cnt"].data_ptr(), k, ids_p, dist_p,
                                b["out_cnt"].data_ptr(), b["scored"].data_ptr())
        else:
            cp = b["cand"].data_ptr() if mode == "gather" else cand_all[bi].data_ptr()
            cx.refine_dev(Q, qp, F32, cp, F32, B, b["sel_ids"].data_ptr(), b["sel_cnt"].data_ptr(), k, ids_p, dist_p,
                          b["out_cnt"].data_ptr(), b["scored"].data_ptr())
        if events is not None:
            events[4].record(stream)
        merge(b, par, stream, si, last)

    def flush(nact=1):
        """gather the batches of a group that is not full yet (end of a timed region: every batch of it is gathered inside it)"""
        if comms is None or MG == 1:
            return
        for si in range(nact):
            n = bufs[si]["nsteps"]
            if n % MG != 0:
                merge(bufs[si], (n // MG) & 1, streams[si], si, True)

    def tick(mode, unfused=False):
        """One step of the 3-deep pipeline on context 0: encode(batch t+2), Route(batch t+1), Refine(batch t) — ONE launch.
        unfused: the same three operations as three stand-alone launches (used for the few steps of the timed region whose
        refinement-scan dispatch carries the kernel-attached events the roofline is read from)."""
        cx, stream, b = ctxs[0], streams[0], bufs[0]
        t = b["tick_no"]
        b["tick_no"] += 1
        bE, bR, bF = (t + 2) % NB, (t + 1) % NB, t % NB
        sE, sR, sF = b["slot"][(t + 2) % 3], b["slot"][(t + 1) % 3], b["slot"][t % 3]
        par, ids_p, dist_p, last = out_slot(b)
        pov = probe_passes[-1]
        if unfused:
            cx.refine_dev(Q, q_all[bF].data_ptr(), F32, cand_all[bF].data_ptr(), F32, B, sF["sel_ids"].data_ptr(), sF["sel_cnt"].data_ptr(), k,
                          ids_p, dist_p, b["out_cnt"].data_ptr(), b["scored"].data_ptr()) if mode == "dense" else \
                cx.refine_store_dev(Q, q_all[bF].data_ptr(), F32, B, sF["sel_ids"].data_ptr(), sF["sel_cnt"].data_ptr(), k, ids_p,
                                    dist_p, b["out_cnt"].data_ptr(), b["scored"].data_ptr())
            cx.route_dev(Q, sR["codes"].data_ptr(), pov, B, B, sR["sel_ids"].data_ptr(), 0, sR["sel_cnt"].data_ptr(), 0, 0)
            cx.encode_dev(Q, q_all[bE].data_ptr(), F32, sE["codes"].data_ptr(), 0, sE["bad"].data_ptr())
        else:
            cx.tick_dev(
                encode=dict(nq=Q, q=q_all[bE].data_ptr(), codes=sE["codes"].data_ptr(), bad=sE["bad"].data_ptr()),
                route=dict(nq=Q, codes=sR["codes"].data_ptr(), limit=B, probe_override=pov, ids=sR["sel_ids"].data_ptr(),
                           count=sR["sel_cnt"].data_ptr(), handover=sR["hov"].data_ptr()),
                refine=dict(nq=Q, q=q_all[bF].data_ptr(), B=B, ids=sF["sel_ids"].data_ptr(), count=sF["sel_cnt"].data_ptr(), k=k,
                            cand=cand_all[bF].data_ptr() if mode == "dense" else None, codes=sF["codes"].data_ptr(), handover=sF["hov"].data_ptr(),
                            probe_override=pov, out_ids=ids_p, out_dist=dist_p, out_count=b["out_cnt"].data_ptr(),
                            scored=b["scored"].data_ptr()))
        merge(b, par, stream, 0, last)

    def tick_prime():
        """fill the pipeline (untimed): encode of batches 0 and 1, Route of batch 0"""
        cx, b = ctxs[0], bufs[0]
        b["tick_no"] = 0
        s0, s1 = b["slot"][0], b["slot"][1]
        cx.encode_dev(Q, q_all[0].data_ptr(), F32, s0["codes"].data_ptr(), 0, s0["bad"].data_ptr())
        cx.route_dev(Q, s0["codes"].data_ptr(), probe_passes[-1], B, B, s0["sel_ids"].data_ptr(), 0, s0["sel_cnt"].data_ptr(), 0, 0)
        cx.encode_dev(Q, q_all[1 % NB].data_ptr(), F32, s1["codes"].data_ptr(), 0, s1["bad"].data_ptr())

    def timed_tick(mode, steps, warmup, with_events=False):
        tick_prime()
        for _ in range(warmup):
            tick(mode)
        barrier()
        every = max(2, steps // max(1, min(8, steps // 40)))
        if with_events:
            ctxs[0].refine_timing_begin(steps, 1)          # only the unfused steps launch a refinement scan of their own
        t_s = time.perf_counter()
        for i in range(steps):
            tick(mode, unfused=with_events and (i % every) == every - 1)
        flush(1)
        ctxs[0].sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t_s
        if use_dist:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        rt = [ctxs[0].refine_timing_end()] if with_events else None
        return el, rt, every

    def barrier():
        for c_ in ctxs:
            c_.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def timed(mode, steps, warmup, nact=1, with_events=False, front=False):
        """`steps` steps over `nact` contexts taking the batches in turn: (elapsed seconds, max over ranks; (dispatches, ms) of the
        refinement-scan launches that carried kernel-attached events; their spacing).  With one context every `every`-th scan
        dispatch carries the events.  With several, kernels of different contexts overlap, so for a SOLO reading every
        `every`-th step first drains all contexts and then runs alone, its scan dispatch carrying the events — inside the
        timed region, and paid for by it."""
        active[0] = nact
        for b_ in bufs:
            b_["nsteps"] = 0
        step_no[0] = 0
        use_front[0] = front and mode == "dense" and len(probe_passes) == 1 and not args.route_counters
        if use_front[0]:
            for si_, (c_, b_) in enumerate(zip(ctxs[:nact], bufs[:nact])):   # fill the pipeline (untimed): the codes of every context's first batch
                b_["front_no"] = 0
                c_.encode_dev(Q, q_all[si_ % NB].data_ptr(), F32, b_["fcodes"][0].data_ptr(), 0, b_["bad"].data_ptr())
        for _ in range(warmup):
            step(mode)
        barrier()
        every = max(2, steps // max(1, min(8, steps // 40)))   # up to eight solo readings, at most one per 40 steps (a drain costs ~2 steps)
        drain_in_region = args.solo_in_region or args.solo_tail <= 0 or shipped
        solo = []
        if with_events and nact == 1:
            ctxs[0].refine_timing_begin(steps, every)
        if with_events and nact > 1:                    # overlapped readings: sampled dispatches of the other contexts, as they run
            for c_ in ctxs[1:nact]:
                c_.refine_timing_begin(steps, max(2, every // 2))
        t_s = time.perf_counter()
        for i in range(steps):
            if with_events and nact > 1 and drain_in_region and (i % every) == every - 1:
                for c_ in ctxs[:nact]:
                    c_.sync()
                ctxs[0].refine_timing_begin(2, 1)        # solo reading: everything drained, this step runs alone on context 0
                step(mode, force_ctx=0)
                solo.append(ctxs[0].refine_timing_end())
            else:
                step(mode)
        flush(nact)
        host_issue_s[0] = time.perf_counter() - t_s      # every launch of the region has been handed to the runtime by now
        for c_ in ctxs[:nact]:
            c_.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t_s
        if use_dist:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        rt = None
        overlapped[0] = None
        if with_events:
            rt = [ctxs[0].refine_timing_end()] if nact == 1 else solo
            if nact > 1:
                ov = [c_.refine_timing_end() for c_ in ctxs[1:nact]]
                on = sum(x for x, _ in ov)
                overlapped[0] = (on, sum(t for _, t in ov) / max(1, on))
        active[0] = 1
        use_front[0] = False
        return el, rt, every

    def solo_readings(mode, count, nact=1, front=False):
        """`count` drained SOLO dispatches of the refinement scan, untimed (after the timed region): the pipeline is set up as in
        timed(), a few steps bring it to its steady state, then every reading drains all contexts and runs ONE step alone on context 0
        whose scan dispatch carries HIP start/stop events.  Consecutive readings take different batches (cold blocks).  Returns the
        per-dispatch milliseconds."""
        active[0] = nact
        for b_ in bufs:
            b_["nsteps"] = 0
        step_no[0] = 0
        use_front[0] = front and mode == "dense" and len(probe_passes) == 1 and not args.route_counters
        if use_front[0]:
            for si_, (c_, b_) in enumerate(zip(ctxs[:nact], bufs[:nact])):
                b_["front_no"] = 0
                c_.encode_dev(Q, q_all[si_ % NB].data_ptr(), F32, b_["fcodes"][0].data_ptr(), 0, b_["bad"].data_ptr())
        for _ in range(2 * nact):
            step(mode)
        out_ms = []
        for _ in range(count):
            for c_ in ctxs[:nact]:
                c_.sync()
            ctxs[0].refine_timing_begin(2, 1)
            step(mode, force_ctx=0)
            ctxs[0].sync()
            n_, ms_ = ctxs[0].refine_timing_end()
            if n_ > 0:
                out_ms.append(ms_ / n_)
        flush(nact)
        barrier()
        active[0] = 1
        use_front[0] = False
        return out_ms

    # ---------------- the timed region ------------------------------------------------------------------------------------
    mode = args.candidates
    use_tick = args.pipeline == "tick" and mode in ("dense", "store") and len(probe_passes) == 1 and not args.route_counters
    # Bring the device to its working state first (clocks, caches, the allocator): an untimed run of the same pipeline.  Without it the
    # first timed region of the process reads ~5 % slower than the same region measured later in the process (seen: front 55.0 us
    # first vs 51.7 us for the three-kernel variant timed after it, although front is the faster one back to back).
    if args.prewarm > 0:
        if use_tick:
            timed_tick(mode, args.prewarm, 2)
        else:
            timed(mode, args.prewarm, 2, nact=nctx, front=args.pipeline == "front")
    if use_tick:
        elapsed, rt, TIMED_EVERY = timed_tick(mode, args.steps, args.warmup, with_events=True)
        tick_fused = ctx.last_tick_fused()
    else:
        elapsed, rt, TIMED_EVERY = timed(mode, args.steps, args.warmup, nact=nctx, with_events=True, front=args.pipeline == "front")
        tick_fused = False
    overlapped_main = overlapped[0]
    host_issue_main = host_issue_s[0]
    # ---------------- what was timed is what is checked: the LAST step each context ran inside the timed region (front_kernel +
    # the scan that finishes PENDING queries, on the clones' own streams) against the plain three-launch path on context 0 for the
    # same batch here, and against the CPU oracle in the cpu_baseline leg ------------------------------------------------------
    timed_check = None
    if args.pipeline == "front" and not use_tick and bufs[0]["last_front"] is not None:
        snaps = []
        for si_, b_ in enumerate(bufs[:nctx]):
            if b_["last_front"] is None:
                continue
            fb_, par_, slot_ = b_["last_front"]
            tk_ = b_["topk"][par_]
            snaps.append(dict(ctx=si_, batch=fb_, ids=tk_.ids[slot_ * Q:(slot_ + 1) * Q].cpu().numpy().copy(),
                              dist=tk_.dist[slot_ * Q:(slot_ + 1) * Q].cpu().numpy().copy(), count=b_["out_cnt"].cpu().numpy().copy(),
                              sel=b_["sel_ids"].cpu().numpy().copy(), sel_cnt=b_["sel_cnt"].cpu().numpy().copy()))
        ok_all = True
        for sn in snaps:
            for b_ in bufs:
                b_["nsteps"] = 0
            step_no[0] = 0
            step(mode, batch=sn["batch"], force_ctx=0)         # plain path: encode, Route (+ hand-back launch), scan on context 0
            barrier()
            r_ids, r_dist = bufs[0]["topk"][0].ids[:Q].cpu().numpy(), bufs[0]["topk"][0].dist[:Q].cpu().numpy()
            r_sel, r_cnt = bufs[0]["sel_ids"].cpu().numpy(), bufs[0]["sel_cnt"].cpu().numpy()
            live = np.arange(B)[None] < r_cnt[:, None]
            same = (np.array_equal(sn["ids"], r_ids) and np.array_equal(sn["dist"], r_dist) and np.array_equal(sn["sel_cnt"], r_cnt)
                    and np.array_equal(np.where(live, sn["sel"], -1), np.where(live, r_sel, -1)))
            sn["same_as_plain"] = bool(same)
            ok_all = ok_all and same
        if not ok_all:
            raise SystemExit("bench: the timed front pipeline's results differ from the plain path: %s"
                             % [(sn["ctx"], sn["batch"], sn["same_as_plain"]) for sn in snaps])
        timed_check = dict(contexts=len(snaps), batches=[sn["batch"] for sn in snaps], same_as_plain_path=True, same_as_oracle=None,
                           note="outputs of the last step every context ran INSIDE the timed region (front_kernel + scan on its own stream), "
                                "compared with the three-launch path on context 0 and, in the cpu_baseline leg, with the CPU oracle")
        timed_snaps = snaps
    # The roofline's readings must not depend on --steps (the driver's 20 steps hold ONE drained dispatch): an untimed tail of
    # drained solo dispatches of the same scan, same pipeline, right behind the timed region (never part of `value`).
    region_n = sum(x for x, _ in rt)
    region_ms = sum(t for _, t in rt)
    tail_ms = []
    if not use_tick and args.solo_tail > 0 and not shipped:
        tail_ms = solo_readings(mode, args.solo_tail, nact=nctx, front=args.pipeline == "front")
    ref_launches = region_n + len(tail_ms)
    ref_ms = (region_ms + sum(tail_ms)) / max(1, ref_launches)     # kernel-attached HIP events, on the context's stream
    solo_stats = (dict(in_region=dict(launches=region_n, avg_launch_ms=round(region_ms / max(1, region_n), 5)),
                       tail=dict(launches=len(tail_ms), median_ms=round(float(np.median(tail_ms)), 5), min_ms=round(min(tail_ms), 5),
                                 max_ms=round(max(tail_ms), 5), mean_ms=round(float(np.mean(tail_ms)), 5),
                                 note="drained solo dispatches behind the timed region (untimed): all contexts idle, one step on context 0, "
                                      "HIP start/stop events attached to its scan dispatch; consecutive readings scan different blocks"))
                  if tail_ms else None)
    ms_per_step = elapsed * 1000.0 / args.steps
    q_job = Qw if args.scaling == "strong" else world * Q
    qps = q_job * args.steps / elapsed

    # ---------------- stage breakdown: a separate, untimed pass with an event after every stage ---------------------------
    nprof = min(args.steps, 20)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(nprof)]
    barrier()
    for i in range(nprof):
        step(mode, evs[i])
    barrier()
    stage_ms = np.array([[evs[i][j].elapsed_time(evs[i][j + 1]) for j in range(4)] for i in range(nprof)])
    st_mean = stage_ms.mean(axis=0)

    # ---------------- encode (north_star: "MFMA utilisation" beside the HBM figure) -------------------------------------------
    # query side: the exact fp64 VALU kernel (sequential chain per lane = Java's dot, bit-exact by construction); index side
    # (Setup, n >= 4096 rows per call): MFMA fp32 GEMM pre-filter with the quantise + bit-pack fused into its epilogue + exact
    # re-check of the pairs it cannot decide.  flops = 2 * rows * d * (T*D*m).
    encode_stage = None
    if rank == 0:
        FP64_VALU_PEAK, F32_MFMA_PEAK = 78.6, 157.3      # TFLOP/s: MI355X fp64 vector (half the fp32 vector rate), fp32 matrix (MI355X_MICROARCH.md)
        fl_q = 2.0 * Q * d * TD * m
        enc_ms = float(st_mean[0])
        nq_e = int(min(n, 1 << 18))
        base_e = ctx.L.fspann_store_dev_ptr(ctx.handle, None)
        codes_e = torch.zeros((nq_e, TD, W), dtype=torch.int64, device=dev)
        bad_e = torch.zeros(nq_e, dtype=torch.int32, device=dev)
        ctx.set_encode_mode(2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            ctx.encode_dev(nq_e, base_e, F32, codes_e.data_ptr(), 0, bad_e.data_ptr())
        ctx.sync()
        e0.record(streams[0])
        for _ in range(5):
            ctx.encode_dev(nq_e, base_e, F32, codes_e.data_ptr(), 0, bad_e.data_ptr())
        e1.record(streams[0])
        ctx.sync()
        mf_ms = e0.elapsed_time(e1) / 5
        rechecked = ctx.last_encode_rechecked()
        codes_m = codes_e.clone()
        ctx.set_encode_mode(1)
        ctx.encode_dev(nq_e, base_e, F32, codes_e.data_ptr(), 0, bad_e.data_ptr())
        ctx.sync()
        ctx.set_encode_mode(0)
        same_codes = bool(torch.equal(codes_m, codes_e))
        if not same_codes:
            raise SystemExit("bench: the MFMA coding path and the exact fp64 kernel disagree on the base rows")
        # north_star's formulation at the QUERY side: the same MFMA pre-filter + exact re-check for one batch of Q queries, stand-alone
        # (fspann_set_encode_mode(2)), next to the exact fp64 kernel (mode 1) — solo launches, events around 50 calls each
        q_enc = {}
        codes_q = torch.zeros((Q, TD, W), dtype=torch.int64, device=dev)
        for mode_e, name_e in ((1, "exact"), (2, "mfma")):
            ctx.set_encode_mode(mode_e)
            for _ in range(5):
                ctx.encode_dev(Q, q_all[0].data_ptr(), F32, codes_q.data_ptr(), 0, bad_e.data_ptr())
            ctx.sync()
            e0.record(streams[0])
            for i in range(50):
                ctx.encode_dev(Q, q_all[i % NB].data_ptr(), F32, codes_q.data_ptr(), 0, bad_e.data_ptr())
            e1.record(streams[0])
            ctx.sync()
            q_enc[name_e] = dict(us_per_call=round(e0.elapsed_time(e1) * 1e3 / 50, 2), rechecked_pairs=int(ctx.last_encode_rechecked()) if mode_e == 2 else None,
                                 codes=codes_q.clone())
        ctx.set_encode_mode(0)
        if not torch.equal(q_enc["exact"]["codes"], q_enc["mfma"]["codes"]):
            raise SystemExit("bench: MFMA and exact coding disagree on a query batch")
        for v_ in q_enc.values():
            del v_["codes"]
        fl_i = 2.0 * nq_e * d * TD * m
        encode_stage = dict(
            query_side_mfma_vs_exact=dict(
                rows=Q, exact_fp64_valu=q_enc["exact"], mfma_prefilter_plus_recheck=q_enc["mfma"], pairs=Q * TD * m, codes_equal=True,
                shipped="exact" if q_enc["exact"]["us_per_call"] <= q_enc["mfma"]["us_per_call"] else "exact (inside front_kernel)",
                note="whole fspann_encode_dev calls back to back on one stream (mode 2 = clear of the code words + encode_mfma_kernel + "
                     "encode_fix_kernel: three dependent launches; 32 x 128 block tile at this size, one MFMA tile per wave); the path wins from ~5e8 multiply-adds per call, "
                     "and in the default pipeline the exact kernel's workgroups ride inside front_kernel beside Route's, where they are off "
                     "the critical path (DESIGN.md §3.1)"),
            query_side=dict(kernel="encode_exact_kernel<float,4> (its workgroups ride inside front_kernel in the default pipeline)", rows=Q,
                            flops_per_launch=fl_q, ms=round(enc_ms, 5), tflops=round(fl_q / (enc_ms * 1e-3) / 1e12, 3), peak=FP64_VALU_PEAK,
                            frac=round(fl_q / (enc_ms * 1e-3) / 1e12 / FP64_VALU_PEAK, 5),
                            bound="fp64 VALU, one sequential dependent chain per lane (Java's dot order): latency of a lone wave, not throughput"),
            index_side=dict(kernel="encode_mfma_kernel<float> (v_mfma_f32_32x32x2_f32, 64 x 256 block tile, quantise + bit-pack in the epilogue) + encode_fix_kernel",
                            rows=nq_e, flops_per_call=fl_i, ms_per_call=round(mf_ms, 4), tflops=round(fl_i / (mf_ms * 1e-3) / 1e12, 2), peak=F32_MFMA_PEAK,
                            frac=round(fl_i / (mf_ms * 1e-3) / 1e12 / F32_MFMA_PEAK, 4), pairs_rechecked_exactly=int(rechecked),
                            pairs=int(nq_e) * TD * m, codes_equal_exact_kernel=same_codes,
                            algorithmic_bytes_per_call=nq_e * (d * 4 + TD * W * 8),
                            note="whole fspann_encode_dev call of 262 144 base rows (clear of the code words, MFMA kernel, re-check kernel), HIP events "
                                 "on the context's stream; every pair the fp32 result cannot decide is recomputed with the exact fp64 chain"))
        del codes_e, codes_m, bad_e

    # ---------------- extra passes (N = 1; reported beside `value`, never as it) -------------------------------------------
    variants, hbm_proof, cfg4_shape, peak_measured, peak_launch_sized = {}, None, None, None, None
    if extras:
        if use_tick or nctx > 1:
            el_s, _, _ = timed(mode, args.steps, max(2, args.warmup))
            variants["serial"] = dict(value=round(Q * args.steps / el_s, 1), unit="queries/s", ms_per_step=round(el_s * 1000.0 / args.steps, 4),
                                      note="ONE context, one stream: the three stages of one batch as three kernels one after the other (encode, Route, "
                                           "Refine); nothing overlaps")
        if args.pipeline == "front" and nctx > 1:
            el_c, _, _ = timed(mode, args.steps, max(2, args.warmup), nact=nctx)
            variants["concurrent"] = dict(value=round(Q * args.steps / el_c, 1), unit="queries/s", ms_per_step=round(el_c * 1000.0 / args.steps, 4),
                                          note="the same contexts with encode, Route and Refine of a batch as three separate kernels per step")
        if mode in ("dense", "store") and len(probe_passes) == 1 and not args.route_counters and not use_tick:
            el_t, _, _ = timed_tick(mode, args.steps, max(2, args.warmup))
            variants["tick"] = dict(value=round(Q * args.steps / el_t, 1), unit="queries/s", ms_per_step=round(el_t * 1000.0 / args.steps, 4),
                                    fused=bool(ctx.last_tick_fused()),
                                    note="ONE stream, three batches in flight: encode(t+2) + Route(t+1) + Refine(t) as one kernel (fspann_tick_dev)")
        if use_tick:
            other = "store" if mode == "dense" else "dense"
            el_o, _, _ = timed_tick(other, args.steps, max(2, args.warmup))
            variants["tick_" + other] = dict(value=round(Q * args.steps / el_o, 1), unit="queries/s", ms_per_step=round(el_o * 1000.0 / args.steps, 4),
                                             note=("trusted-HBM variant of the same pipeline: Refine reads plaintext rows of an HBM-resident store by id; "
                                                   "production keeps decrypt on the host, so this is NOT the reference's boundary") if other == "store"
                                             else "the same pipeline over resident [Q][B][d] blocks (SURVEY 8d kernel path)")
        for vm, note in (("store", "trusted-HBM variant: Refine reads plaintext rows of an HBM-resident store by id (one library call per step); "
                                   "production keeps decrypt on the host, so this is NOT the reference's boundary"),
                         ("gather", "the [Q][B][d] block is packed inside the step by a gather kernel (device stand-in for load + decrypt), then scanned"),
                         ("dense", "the [Q][B][d] block is resident before the step (SURVEY 8d kernel path)")):
            if vm == mode and not use_tick:
                continue
            el_v, _, _ = timed(vm, args.steps, max(2, args.warmup), nact=nctx)
            variants[vm] = dict(value=round(Q * args.steps / el_v, 1), unit="queries/s", ms_per_step=round(el_v * 1000.0 / args.steps, 4), note=note)

    # ---------------- the operator surface's OWN call pattern (N = 1) ------------------------------------------------------------
    # QueryService.search(QueryToken) is called once per query by the reference's loop (ForwardSecureANNSystem.java:636-748), so the
    # Java adapter (GpuQueryServiceImpl.search) issues, per query and through HOST pointers: fspann_route(nq = 1, limit = B) ->
    # the host's load + decrypt loop -> fspann_refine(nq = 1, fp64 rows) — the token's codes come from fspann_encode(nq = 1) at
    # token creation.  Timed here exactly like that through the same C entry points (ctypes instead of JNI; the plaintext rows are
    # copied out of the base array where the JVM would decrypt: that copy is reported, not counted), next to the batched mirror
    # searchBatch (one fspann_route + one fspann_refine for all queries of a batch).
    operator_surface = None
    if rank == 0 and world == 1 and not args.no_extras:
        nsamp = 256 if B <= 1024 else 64
        qs_ = Qall[0][:nsamp].astype(np.float64)
        lat = {k_: [] for k_ in ("encode", "route", "host_rows", "refine", "refine_pageable")}
        hb1 = ctx.host_buffer((1, B, d), np.float64)       # the context's pinned block: where the adapter packs the decrypted rows
        for pinned_rows in (False, True):                   # (first with the rows in ordinary memory, as a caller unaware of the block would)
            for i in range(8 + nsamp):
                q1 = qs_[i % nsamp][None]
                t0_ = time.perf_counter()
                c1 = ctx.encode(q1)
                t1_ = time.perf_counter()
                r1 = ctx.route(c1, probe_override=probe_passes[-1], limit=B, counters=False)
                t2_ = time.perf_counter()
                cN = int(r1["count"][0])
                if pinned_rows:
                    hb1[0, :cN] = X[r1["ids"][0, :cN]]
                    rows = hb1[:, :cN]
                else:
                    rows = X[r1["ids"][0, :cN]].astype(np.float64)[None]
                t3_ = time.perf_counter()
                if cN > 0:
                    ctx.refine(q1, rows, np.arange(cN, dtype=np.int32)[None], np.array([cN], np.int32), k)
                t4_ = time.perf_counter()
                if i >= 8 and pinned_rows:
                    for k_, v_ in (("encode", t1_ - t0_), ("route", t2_ - t1_), ("host_rows", t3_ - t2_), ("refine", t4_ - t3_)):
                        lat[k_].append(v_ * 1e6)
                elif i >= 8:
                    lat["refine_pageable"].append((t4_ - t3_) * 1e6)
        gpu_calls = np.array(lat["route"]) + np.array(lat["refine"])
        # the batched mirror: the same three calls for all nsamp queries at once
        tb0 = time.perf_counter()
        cB = ctx.encode(qs_)
        tb1 = time.perf_counter()
        rB = ctx.route(cB, probe_override=probe_passes[-1], limit=B, counters=False)
        tb2 = time.perf_counter()
        cntB = rB["count"].astype(np.int32)
        rowsB = ctx.host_buffer((nsamp, B, d), np.float64)
        for i in range(nsamp):
            rowsB[i, :cntB[i]] = X[rB["ids"][i, :cntB[i]]]
        tb3 = time.perf_counter()
        ctx.refine(qs_, rowsB, np.tile(np.arange(B, dtype=np.int32), (nsamp, 1)), cntB, k)
        tb4 = time.perf_counter()
        del rowsB

        def pct(a, p_):
            return round(float(np.percentile(np.asarray(a), p_)), 1)
        operator_surface = dict(
            per_query=dict(queries=nsamp, unit="us per query",
                           route_plus_refine=dict(p50=pct(gpu_calls, 50), p99=pct(gpu_calls, 99), mean=round(float(gpu_calls.mean()), 1)),
                           encode=dict(p50=pct(lat["encode"], 50), p99=pct(lat["encode"], 99)),
                           route=dict(p50=pct(lat["route"], 50), p99=pct(lat["route"], 99)),
                           refine_f64_rows=dict(p50=pct(lat["refine"], 50), p99=pct(lat["refine"], 99)),
                           refine_f64_rows_from_pageable_memory=dict(p50=pct(lat["refine_pageable"], 50), p99=pct(lat["refine_pageable"], 99)),
                           host_rows_copy_not_counted=dict(p50=pct(lat["host_rows"], 50)),
                           queries_per_s=round(1e6 / float(gpu_calls.mean()), 1),
                           note="GpuQueryServiceImpl.search's pattern: fspann_route(nq = 1, limit = B, host pointers) then fspann_refine(nq = 1, fp64 "
                                "rows packed into the context's pinned block, fspann_host_buffer: one DMA of B x d x 8 bytes); calls of a handful of "
                                "queries read their small arguments from, and write their results into, mapped pinned memory (no copy commands); "
                                "synchronous, one query in flight, Python ctypes call overhead included"),
            batched=dict(queries=nsamp, unit="us per query", encode=round((tb1 - tb0) * 1e6 / nsamp, 2), route=round((tb2 - tb1) * 1e6 / nsamp, 2),
                         refine_f64_rows=round((tb4 - tb3) * 1e6 / nsamp, 2), host_rows_copy_not_counted=round((tb3 - tb2) * 1e6 / nsamp, 2),
                         queries_per_s=round(nsamp / ((tb2 - tb1) + (tb4 - tb3)), 1),
                         note="searchBatch (GpuQueryServiceImpl.searchBatch / operators.QueryServiceImpl.search_batch): ONE fspann_route and ONE "
                              "fspann_refine(fp64 rows) for the whole batch through the same host-pointer entry points (PCIe-inclusive)"))

    # ---------------- end to end (N = 1): the native host candidate pipeline — Route on the GPU, AES-256-GCM open of F_q's records
    # on the host cores into pinned staging, H2D, Refine — three batches in flight (fspann_pipeline_*) ------------
    end_to_end = None
    if extras:
        from fspann_amd import hostpipe
        try:
            nthr = len(os.sched_getaffinity(0))
        except AttributeError:
            nthr = os.cpu_count() or 1
        nthr = min(nthr, args.host_threads) if args.host_threads > 0 else nthr
        with hostpipe.PointStore(n, d) as ps:
            t_e = time.perf_counter()
            ps.encrypt(X, threads=nthr)
            t_e = time.perf_counter() - t_e
            with hostpipe.Pipeline(ctx, ps, Q, B, k, host_threads=nthr) as pl:
                nbe = 10

                def run_e2e(nb_):
                    got = []
                    for i in range(nb_):
                        pl.submit(Qall[i % NB])
                        if pl.in_flight == 4:
                            got.append(pl.collect())
                    while pl.in_flight:
                        got.append(pl.collect())
                    return got
                run_e2e(2)
                t_p = time.perf_counter()
                got = run_e2e(nbe)
                t_p = time.perf_counter() - t_p
                st = pl.stats()
            e2e_first = got[0]
        end_to_end = dict(value=round(Q * nbe / t_p, 1), unit="queries/s", batches=nbe, host_threads=nthr,
                          ms_per_batch=round(t_p * 1000.0 / nbe, 2), stage_ms=dict(route=round(st["route_ms"], 3), decrypt=round(st["decrypt_ms"], 3),
                                                                                   h2d_refine=round(st["refine_ms"], 3)),
                          candidate_bytes_per_batch=Q * B * d * 4, records_opened_per_batch=Q * B,
                          store_encrypt_s=round(t_e, 2),
                          note="fspann_pipeline: Route (GPU) | AES-256-GCM open of 262 144 records per batch on the host threads into pinned staging | "
                               "H2D + Refine (GPU), four batches in flight; record format, AAD and key derivation are the reference's "
                               "(AesGcmCryptoService.java:55-166, EncryptedPoint.java:80-83, KeyManager.java:221-237); host-bound by design")

    # ... and what ONE host thread pays per open against the bare cipher (tools/micro/open_bench.cpp: the store's own header built
    # with g++ on this box; a report, skipped when there is no compiler)
    if end_to_end is not None:
        try:
            import re
            import subprocess
            import tempfile
            exe = os.path.join(tempfile.gettempdir(), "fspann_open_bench_%d" % os.getpid())
            subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tools", "micro", "open_bench.cpp"), "-ldl", "-o", exe],
                           check=True, capture_output=True, timeout=120)
            ob = subprocess.run([exe, str(Q * B), str(end_to_end["host_threads"])], check=True, capture_output=True, text=True, timeout=120).stdout
            os.unlink(exe)
            bare = float(re.search(r"hot: ([0-9.]+) us per message", ob).group(1))
            one = float(re.search(r"pointstore_open_one, 1 thread, scattered: ([0-9.]+) us", ob).group(1))
            many = [float(x) for x in re.findall(r"pointstore_open_batch<float>, \d+ threads: ([0-9.]+) ms", ob)]
            end_to_end["per_open"] = dict(
                bare_cipher_us=bare, open_one_thread_us=one, ratio_to_bare_cipher=round(one / bare, 2),
                opens_per_s_per_core=round(1e6 / one, 1),
                batch_ms=min(many) if many else None,
                cores_worth_of_the_thread_pool=round((Q * B / (min(many) * 1e-3)) / (1e6 / one), 2) if many else None,
                parts_us={k_: float(v_) for k_, v_ in re.findall(r"^  (.+?)\s{2,}([0-9.]+) us", ob, flags=re.M)},
                note="one thread: EVP AES-256-GCM over one hot 1 040-byte record (bare cipher) vs pointstore_open_one over scattered records "
                     "(version check, snapshot, AAD, cipher, tag, big-endian fp64 decode); the thread pool's rate over the one-thread rate = how "
                     "many cores' worth of CPU the box gives this process")
        except Exception as e:
            end_to_end["per_open"] = dict(error=str(e)[:200])

    # one more (untimed) step of batch 0 on context 0: its results are what recall and the CPU baseline are checked on
    flagged_in_timed_runs = sum(int(c_.unmodelled_queries()) for c_ in ctxs)     # (read + reset: the timed steps leave such queries empty)
    for b_ in bufs:
        b_["nsteps"] = 0
    step_no[0] = 0
    barrier()
    if use_tick:       # batch 0 is refined by tick 0 (its encode and Route were done by tick_prime)
        tick_prime()
        tick(mode)
    else:
        step(mode, batch=0)   # (active[0] == 1: context 0)
    flush(1)
    barrier()
    # Queries whose bestScore map treeifies a bin (count -1 after Route; ~0.3 % at the shipped profiles, ~2e-9 at the headline
    # workload) are finished by the library's host model — outside the timed region, which leaves them empty: reported below.
    treeified_batch0 = 0
    if not use_tick and ctx.unmodelled_queries(reset=False) != 0:
        b0_ = bufs[0]
        treeified_batch0 = ctx.route_resolve_dev(Q, b0_["codes"].data_ptr(), probe_passes[-1], B, B, b0_["sel_ids"].data_ptr(), 0, b0_["sel_cnt"].data_ptr())
        tk0 = b0_["topk"][0]
        if mode == "store":
            ctx.refine_store_dev(Q, q_all[0].data_ptr(), F32, B, b0_["sel_ids"].data_ptr(), b0_["sel_cnt"].data_ptr(), k, tk0.ids.data_ptr(), tk0.dist.data_ptr(),
                                 b0_["out_cnt"].data_ptr(), b0_["scored"].data_ptr())
        else:
            cp0 = b0_["cand"].data_ptr() if mode == "gather" else cand_all[0].data_ptr()
            if mode == "gather":
                ctx.store_gather_dev(Q, b0_["sel_ids"].data_ptr(), b0_["sel_cnt"].data_ptr(), B, cp0)
            ctx.refine_dev(Q, q_all[0].data_ptr(), F32, cp0, F32, B, b0_["sel_ids"].data_ptr(), b0_["sel_cnt"].data_ptr(), k, tk0.ids.data_ptr(),
                           tk0.dist.data_ptr(), b0_["out_cnt"].data_ptr(), b0_["scored"].data_ptr())
        ctx.sync()
    out_ids, out_dist = bufs[0]["topk"][0].ids[:Q], bufs[0]["topk"][0].dist[:Q]      # first batch of the first group
    got_ids, got_dist = out_ids.cpu().numpy(), out_dist.cpu().numpy()
    if ctx.unmodelled_queries() != 0:
        raise SystemExit("bench: a query stayed unmodelled (equal hashCodes of non-decimal ids inside a treeified bin)")
    if end_to_end is not None:      # same batch through the decrypting pipeline: same answer
        end_to_end["matches_kernel_path"] = bool(np.array_equal(e2e_first["ids"], got_ids) and np.array_equal(e2e_first["dist"], got_dist))
        if not end_to_end["matches_kernel_path"]:
            raise SystemExit("bench: the end-to-end pipeline and the kernel path disagree on batch 0")

    # ---------------- roofline of the refinement scan (north_star's HBM-bound kernel) ---------------------------------------
    # algorithmic bytes per launch: Q * (B*d*4 + d*4 + k*8)
    ref_bytes = Q * (B * d * 4 + d * 4 + k * 8)
    achieved = ref_bytes / (ref_ms * 1e-3) / 1e9 if ref_ms > 0 else 0.0      # (0: no dispatch carried events — experiments only)
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "refine_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath)).get(mode, {})
            if tj.get("workload") == args.workload and tj.get("Q") == Q:
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), tj.get("source")
        except Exception:
            traffic = None
    if extras:
        peak_measured = round(ctx.hbm_read_peak(1 << 32, 5), 1)      # pure 16-byte-load kernel over 4 GiB, best of 5
        # the same pure-load kernel when one launch reads only as many bytes as one refinement scan does (cold windows of 4 GiB)
        peak_launch_sized = round(ctx.hbm_read_window(1 << 32, (Q * (B * d * 4 + d * 4 + k * 8) + 15) // 16 * 16, 24), 1)
        # (a) the same scan reading 1024 x 256 random rows of an 8 M x 128 (4.1 GB) store: 134 MB per launch out of 4.1 GB
        NS = 8_000_000
        big = torch.randn((NS, d), dtype=torch.float32, device=dev)
        rid = torch.randint(0, NS, (8, Q, B), dtype=torch.int32, device=dev)
        cnt = torch.full((Q,), B, dtype=torch.int32, device=dev)
        b0 = bufs[0]
        torch.cuda.synchronize()
        with pkg.FspannContext(cfg, local_rank) as cp:         # Refine needs no index: a bare context over the big store
            cp.store_attach_dev(NS, big.data_ptr(), F32)

            def rs(i):
                cp.refine_store_dev(Q, q_all[0].data_ptr(), F32, B, rid[i % 8].data_ptr(), cnt.data_ptr(), k, b0["topk"][1].ids.data_ptr(),
                                    b0["topk"][1].dist.data_ptr(), b0["out_cnt"].data_ptr(), b0["scored"].data_ptr())
            for i in range(4):
                rs(i)
            cp.sync()
            cp.refine_timing_begin(40, 1)
            for i in range(40):
                rs(i)
            nl, tms = cp.refine_timing_end()
        del big, rid
        hp_ms = tms / max(1, nl)
        hbm_proof = dict(kernel="refine_stream_kernel<float,float,32,true>", store_rows=NS, store_bytes=NS * d * 4, launches=nl,
                         avg_launch_ms=round(hp_ms, 5), achieved=round(ref_bytes / (hp_ms * 1e-3) / 1e9, 1),
                         frac=round(ref_bytes / (hp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         note="1024 x 256 uniformly random rows of a 4.1 GB store per launch, 8 id sets cycled: at most 6 % of a launch's rows "
                              "can sit in the 256 MiB Infinity Cache")
        # (b) BASELINE config #4's refine shape: d = 768, B = 1024 -> four 256-row chunks per query + refine_merge_kernel
        d4, B4 = 768, 1024
        cfg4 = pkg.PaperRuntimeConfig(tables=1, divisions=1, m=8, lambda_=2, dim=d4, refinement_limit=B4)
        with pkg.FspannContext(cfg4, local_rank) as c4:
            cand4 = torch.randn((2, Q, B4, d4), dtype=torch.float32, device=dev)          # 2 x 3.2 GB
            q4 = torch.randn((Q, d4), dtype=torch.float32, device=dev)
            ids4 = torch.arange(Q * B4, dtype=torch.int32, device=dev).view(Q, B4)
            cnt4 = torch.full((Q,), B4, dtype=torch.int32, device=dev)
            oi4 = torch.zeros((Q, k), dtype=torch.int32, device=dev)
            od4 = torch.zeros((Q, k), dtype=torch.float64, device=dev)
            oc4 = torch.zeros(Q, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            s4 = torch.cuda.ExternalStream(c4.stream, device=dev)

            def r4(i):
                c4.refine_dev(Q, q4.data_ptr(), F32, cand4[i & 1].data_ptr(), F32, B4, ids4.data_ptr(), cnt4.data_ptr(), k, oi4.data_ptr(),
                              od4.data_ptr(), oc4.data_ptr(), 0)
            for i in range(2):
                r4(i)
            c4.sync()
            c4.refine_timing_begin(12, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s4)
            for i in range(12):
                r4(i)
            e1.record(s4)
            nl4, tms4 = c4.refine_timing_end()
            bytes4 = Q * (B4 * d4 * 4 + d4 * 4 + k * 8)
            scan_ms, call_ms = tms4 / max(1, nl4), e0.elapsed_time(e1) / 12
            cfg4_shape = dict(kernel="refine_stream_kernel<float,float,32,false> over 4 chunks per query + refine_merge_kernel", Q=Q, B=B4, dim=d4,
                              algorithmic_bytes_per_launch=bytes4, scan_launch_ms=round(scan_ms, 4), scan_plus_merge_ms=round(call_ms, 4),
                              achieved=round(bytes4 / (scan_ms * 1e-3) / 1e9, 1), frac=round(bytes4 / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              note="two 3.2 GB blocks alternate; events attached to the scan kernel; scan_plus_merge_ms = whole fspann_refine_dev call")
            del cand4
        # (c) fp64 rows, fp64 query — what the JVM hands over (double[] from decryptFromPoint): refine_stream_kernel<double,double,16,false>,
        # algorithmic bytes = Q * (B*d*8 + d*8 + k*8); 8 distinct blocks are cycled (2.1 GB at config #2)
        NB64 = 8
        cand64 = torch.randn((NB64, Q, B, d), dtype=torch.float64, device=dev)
        q64 = q_all[0].double()
        ids64 = torch.arange(Q * B, dtype=torch.int32, device=dev).view(Q, B)
        cnt64 = torch.full((Q,), B, dtype=torch.int32, device=dev)
        b0 = bufs[0]
        torch.cuda.synchronize()

        def r64(i):
            ctx.refine_dev(Q, q64.data_ptr(), pkg._native.F64, cand64[i % NB64].data_ptr(), pkg._native.F64, B, ids64.data_ptr(), cnt64.data_ptr(), k,
                           b0["topk"][1].ids.data_ptr(), b0["topk"][1].dist.data_ptr(), b0["out_cnt"].data_ptr(), b0["scored"].data_ptr())
        for i in range(4):
            r64(i)
        ctx.sync()
        ctx.refine_timing_begin(32, 1)
        for i in range(32):
            r64(i)
            ctx.sync()                        # solo dispatches
        nl64, tms64 = ctx.refine_timing_end()
        bytes64 = Q * (B * d * 8 + d * 8 + k * 8)
        ms64 = tms64 / max(1, nl64)
        f64_rows = dict(kernel="refine_stream_kernel<double,double,16,false>", algorithmic_bytes_per_launch=bytes64, launches=nl64,
                        avg_launch_ms=round(ms64, 5), achieved=round(bytes64 / (ms64 * 1e-3) / 1e9, 1),
                        frac=round(bytes64 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        note="the scan over fp64 candidate rows and an fp64 query (the dtype GpuQueryServiceImpl hands over), device-resident "
                             "blocks, solo dispatches with kernel-attached events")
        del cand64, q64
    else:
        f64_rows = None
    kname = "refine_stream_kernel<float,float,32,%s>" % ("true" if mode == "store" else "false")
    if mode != "store" and B > 256 and 32 < k <= 128:
        kname = ("refine_stream_kernel<float,float,32,false,true> (a workgroup walks a run of consecutive chunks of one query and keeps its best k "
                 "in LDS: one list per run, no merge kernel when a run is the whole query)")
    if args.pipeline == "front" and not use_tick and mode == "dense" and B <= 256:
        kname = "refine_stream_fix_kernel<false> (the streaming scan whose workgroups first finish queries the bounded select handed over)"
    roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, peak_spec=HBM_PEAK_GBS,
                    peak_measured=peak_measured, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                    frac_of_measured=round(achieved / peak_measured, 4) if peak_measured else None,
                    peak_launch_sized=peak_launch_sized,
                    frac_of_launch_sized=round(achieved / peak_launch_sized, 4) if peak_launch_sized else None,
                    traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=ref_bytes, avg_launch_ms=round(ref_ms, 5), launches=ref_launches, solo=solo_stats,
                    working_set_bytes=(NB * Q * B * d * 4) if mode == "dense" else n * d * 4,
                    timing=("every %d-th step of the timed region runs its three stages as stand-alone kernels instead of the shared one; the "
                            "refinement-scan dispatch of those steps carries HIP start/stop events (hipExtLaunchKernel, on the context's stream)"
                            if use_tick else (("every %d-th step of the timed region first drains all contexts and then runs alone; its refinement-scan "
                                               "dispatch carries HIP start/stop events (hipExtLaunchKernel, on its context's stream): a SOLO duration, "
                                               "measured inside the timed region") if (args.solo_in_region or args.solo_tail <= 0 or shipped) else
                                              ("SOLO durations from the untimed tail behind the timed region (roofline.solo.tail: drained dispatches of the same "
                                               "scan in the same pipeline, HIP start/stop events attached to each, hipExtLaunchKernel on the context's stream); "
                                               "inside the timed region — which holds exactly --steps pipeline steps, no drain — every %d-th scan dispatch of the "
                                               "other contexts carries events too: roofline.overlapped")) if nctx > 1 else
                            "HIP start/stop events attached to every %d-th refinement-scan dispatch of the timed region "
                            "(hipExtLaunchKernel, on the context's stream)") % TIMED_EVERY,
                    bracket_ms=round(float(st_mean[3]), 5),
                    overlapped=(dict(launches=overlapped_main[0], avg_launch_ms=round(overlapped_main[1], 5),
                                     achieved=round(ref_bytes / (overlapped_main[1] * 1e-3) / 1e9, 1),
                                     note="the same kernel's sampled dispatches while kernels of the other contexts run beside it (what a kernel "
                                          "trace of this command averages over): it shares HBM then, so this is not a roofline of the kernel")
                                if overlapped_main and overlapped_main[1] > 0 else None),
                    hbm_proof=hbm_proof, cfg4_shape=cfg4_shape, f64_rows=f64_rows)

    # Route (probe + select) is bound by dependent L2 rounds and LDS atomics, not by HBM or MFMA; its algorithmic bytes
    # (per (t,d) search + rep/id-range fetch + P*S ids) are reported for scale.
    P_, S_ = P_eff, 64
    nparts = (n + S_ - 1) // S_
    levels = max(1, int(np.ceil(np.log(max(nparts, 2)) / np.log(16))))
    route_bytes = Q * (TD * (levels * 16 * 16 + (2 * P_ - 1) * (8 * W + 8) + P_ * S_ * 4) + B * 4)
    route_ms = float(st_mean[1])
    rinfo = ctx.last_route_info()
    route_info = dict(kernels=("route_select_lazy_kernel<256> with the probe fused in (bounded select; %d of %d queries handed to the full select)"
                               % (rinfo["overflowed"], Q)) if rinfo["lazy"] else "route_probe_kernel + route_select_kernel<true,512>",
                      bound="L2 latency + LDS atomics (integer)", avg_ms=round(route_ms, 5), algorithmic_bytes_per_launch=int(route_bytes),
                      achieved_GBs=round(route_bytes / (route_ms * 1e-3) / 1e9, 1))

    # ---------------- recall@10 + distance ratio vs exact kNN of the data set (reported, never assumed) ------------------
    recall = ratio = None
    gt_ms = None
    if rank == 0:
        # exact ground truth on the GPU with the reference's own arithmetic and tie rule (fspann_groundtruth_dev =
        # GroundtruthPrecompute.run), recall@k / distance ratio@k as ForwardSecureANNSystem.computeMetricsAtK defines them
        nv = q_live if args.scaling == "strong" else Q
        base_ptr = ctx.L.fspann_store_dev_ptr(ctx.handle, None)            # the fp32 rows already resident in HBM
        gt_d = torch.zeros((Q, k), dtype=torch.int32, device=dev)
        rec_d = torch.zeros(Q, dtype=torch.float64, device=dev)
        rat_d = torch.zeros(Q, dtype=torch.float64, device=dev)
        ann_d = out_ids.contiguous()
        torch.cuda.synchronize()
        t_g = time.perf_counter()
        ctx.groundtruth_dev(n, base_ptr, Q, q_all[0].data_ptr(), d, k, gt_d.data_ptr())
        ctx.sync()
        gt_ms = (time.perf_counter() - t_g) * 1e3
        ctx.eval_metrics_dev(n, base_ptr, Q, q_all[0].data_ptr(), d, k, ann_d.data_ptr(), k, bufs[0]["out_cnt"].data_ptr(), gt_d.data_ptr(), k,
                             rec_d.data_ptr(), rat_d.data_ptr())
        ctx.sync()
        rec_h, rat_h = rec_d.cpu().numpy()[:nv], rat_d.cpu().numpy()[:nv]
        recall = float(rec_h.mean())
        ratio = float(np.nanmean(rat_h)) if np.isfinite(rat_h).any() else None

    # ---------------- recall / ratio as the candidate budget grows (untimed; README.md:299-303 quotes B = 22 000 for 0.88 on SIFT1M) --
    # B beyond T*D*probes*blockSize cannot be reached with the default 5 probes: the sweep raises the probe override with B.
    recall_sweep = None
    if rank == 0 and extras and recall is not None:
        recall_sweep = []
        S_blk = 64
        for Bs in (256, 6000, 22000):
            pov = -1 if Bs <= 256 else min(64, -(-Bs // (TD * S_blk)) + 2)          # enough probed partitions for Bs distinct ids
            try:
                cfg_s = pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d, seed=13, refinement_limit=Bs,
                                               max_global_candidates=max(20000, Bs))
                with pkg.FspannContext(cfg_s, local_rank) as cs:
                    cs.set_gfunctions(*ctx.get_gfunctions())
                    cs.set_id_meta(n)
                    for td in range(TD):
                        cs.set_index(td, **ctx.get_index(td))
                    cs.finalize()
                    cs.store_attach_dev(n, ctx.L.fspann_store_dev_ptr(ctx.handle, None), F32)
                    s_ids = torch.full((Q, Bs), -1, dtype=torch.int32, device=dev)
                    s_cnt = torch.zeros(Q, dtype=torch.int32, device=dev)
                    o_ids = torch.zeros((Q, k), dtype=torch.int32, device=dev)
                    o_dst = torch.zeros((Q, k), dtype=torch.float64, device=dev)
                    o_cnt = torch.zeros(Q, dtype=torch.int32, device=dev)
                    o_sc = torch.zeros(Q, dtype=torch.int32, device=dev)
                    o_bad = torch.zeros(Q, dtype=torch.int32, device=dev)
                    torch.cuda.synchronize()
                    t_s = time.perf_counter()
                    cs.search_store_dev(Q, q_all[0].data_ptr(), F32, pov, Bs, k, o_ids.data_ptr(), o_dst.data_ptr(), o_cnt.data_ptr(), o_sc.data_ptr(),
                                        s_ids.data_ptr(), s_cnt.data_ptr(), o_bad.data_ptr())
                    cs.sync()
                    ms_s = (time.perf_counter() - t_s) * 1e3
                    tree_s = cs.search_store_finish_dev(Q, q_all[0].data_ptr(), F32, pov, Bs, k, o_ids.data_ptr(), o_dst.data_ptr(), o_cnt.data_ptr(),
                                                        o_sc.data_ptr(), s_ids.data_ptr(), s_cnt.data_ptr())     # host model for treeified bins
                    cs.eval_metrics_dev(n, base_ptr, Q, q_all[0].data_ptr(), d, k, o_ids.data_ptr(), k, o_cnt.data_ptr(), gt_d.data_ptr(), k,
                                        rec_d.data_ptr(), rat_d.data_ptr())
                    cs.sync()
                    rat_s = rat_d.cpu().numpy()[:nv]
                    recall_sweep.append(dict(B=Bs, probes=int(cs.effective_probes(pov)), recall_at_10=round(float(rec_d.cpu().numpy()[:nv].mean()), 5),
                                             distance_ratio_at_10=round(float(np.nanmean(rat_s)), 5) if np.isfinite(rat_s).any() else None,
                                             scored_mean=round(float(o_sc.float().mean().item()), 1), ms_first_call=round(ms_s, 2),
                                             treeified_finished_on_host=int(tree_s), unmodelled=int(cs.unmodelled_queries())))
                    del s_ids
            except Exception as e:      # the sweep is a report, never a reason to lose the bench line
                recall_sweep.append(dict(B=Bs, error=str(e)[:200]))

    # ---------------- CPU baseline: the oracle on this box's host cores (rank 0, N = 1 only) --------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        O = graft.load_oracle()
        o = O.Oracle(T, D, m, lam, d, refinement_limit=B, max_global_candidates=int(wl.get("hard_cap", 20000)),
                     probe_override=P_cfg if P_cfg > 0 else -1)
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
        ns = min(args.cpu_sample if not shipped else min(args.cpu_sample, 128), Q)      # ~6 ms per query at B = 22 000
        qs = Qall[0][:ns].astype(np.float64)
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
        same = bool(np.array_equal(ref["ids"], got_ids[:ns]) and np.array_equal(ref["dist"], got_dist[:ns]))
        if o.unmodelled:
            raise SystemExit("bench: a HashMap bin treeified in the oracle at this workload: the checker has no pinned order")
        if not (same and index_same):
            raise SystemExit(f"bench: GPU results differ from the CPU oracle (index_same={index_same}, results_same={same})")
        if timed_check is not None:      # the timed launches themselves, every context, all queries
            try:
                nthr_o = len(os.sched_getaffinity(0))
            except AttributeError:
                nthr_o = os.cpu_count() or 1
            for sn in timed_snaps:
                ro = o.search(Qall[sn["batch"]].astype(np.float64), k, threads=nthr_o)
                okq = (np.array_equal(ro["ids"], sn["ids"]) and np.array_equal(ro["dist"], sn["dist"]) and np.array_equal(ro["sel_count"], sn["sel_cnt"])
                       and np.array_equal(ro["sel"][:, :B], np.where(np.arange(B)[None] < sn["sel_cnt"][:, None], sn["sel"], -1)))
                if not okq or o.unmodelled:
                    raise SystemExit(f"bench: timed front step of context {sn['ctx']} (batch {sn['batch']}) differs from the CPU oracle")
            timed_check["same_as_oracle"] = True
        cpu = dict(value=round(done / cpu_s, 1), unit="queries/s", cores=1, kind="port",
                   sample=f"{ns} queries x {reps} passes of the same batch (encode + Route + Refine on plaintext, "
                          f"no AES/RocksDB), C++ oracle single thread; host has {os.cpu_count()} logical cores",
                   matches_gpu=same, index_matches_gpu=bool(index_same), oracle_unmodelled=bool(o.unmodelled),
                   oracle_index_build_s=round(t_ob, 1))
        # the same port over ALL host cores this process may use (queries are independent: threads over queries, encode too)
        try:
            nthr = len(os.sched_getaffinity(0))
        except AttributeError:
            nthr = os.cpu_count() or 1
        if nthr > 1:
            qall64 = Qall[:4].reshape(-1, d).astype(np.float64)       # 4 batches: enough work per thread
            o.search(qall64[:256], k, threads=nthr)
            t2 = time.perf_counter()
            reps2, done2 = 0, 0
            while True:
                o.search(qall64, k, threads=nthr)                      # encode + Route + Refine, all threaded over queries
                reps2 += 1
                done2 += len(qall64)
                if time.perf_counter() - t2 > 5.0 or reps2 >= 200:
                    break
            v_all = done2 / (time.perf_counter() - t2)
            cpu["all_cores"] = dict(value=round(v_all, 1), unit="queries/s", threads=nthr,
                                    cores_worth=round(v_all / max(cpu["value"], 1e-9), 2),
                                    note="the same port with one thread per logical CPU the process may be scheduled on; the box gives this process only "
                                         "a share of them: `cores_worth` = this rate / the single-thread rate = how many cores' worth of work it actually got")

    if use_dist and comms is not None:
        # merged result = every rank's top-k in rank order.  EVERY rank checks EVERY slice: the buffer the path's own collective
        # filled against an independent torch.distributed all-gather of the same local buffers, then all ranks agree.
        gat, loc = bufs[0]["gathered"][0], bufs[0]["topk"][0]
        ref_raw = torch.zeros_like(gat.raw)
        torch.cuda.synchronize()
        dist.all_gather_into_tensor(ref_raw, loc.raw)
        torch.cuda.synchronize()
        g_ids, g_dist = gat.split()
        own = torch.equal(g_ids[rank * Q * MG: rank * Q * MG + Q], out_ids) and torch.equal(g_dist[rank * Q * MG: rank * Q * MG + Q], out_dist)
        okf = torch.tensor([1 if (own and torch.equal(ref_raw, gat.raw)) else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if int(okf.item()) != 1:
            raise SystemExit(f"bench: rank {rank}: the gathered top-k differs from an independent all-gather of the ranks' buffers")
        ranks_verified = True
    else:
        ranks_verified = None
    # ---------------- the reference's SHIPPED profiles, as children of the default command (rank 0, N = 1) -------------------------
    # config_sift1m.json:44-56,116-128 (SIFT_P4_FAST, SIFT_P10_HIGH; k = 100): the only configurations BASELINE.md holds reference
    # numbers for.  Each runs as its own `bench.py --workload ...` process AFTER everything above (this process is idle meanwhile and
    # keeps only its buffers), on SIFT-like synthetic data (integers 0..255 of intrinsic dimension 16, make_data: recall there is in the range
    # the reference publishes for real SIFT1M, quoted beside it); its line is summarised under `extra.shipped_profiles`.
    extra = None
    if rank == 0 and world == 1 and extras and not args.no_shipped:
        import subprocess
        torch.cuda.synchronize()
        shipped_out = {}
        # what the reference publishes for these two profiles (real SIFT1M, Xeon E5-2630 v4, one query thread, end to end incl.
        # AES-GCM and RocksDB): context for the recall / ratio reached here on synthetic data, not a baseline for `value`
        ref_pub = {"sift1m_P4_FAST": dict(recall_at_100=0.5506, distance_ratio=1.0276, art_ms=1429.8, source="fsp-anns-parent/logs/New Results:27-30"),
                   "sift1m_P10_HIGH": dict(recall_at_100=0.7714, distance_ratio=1.0097, art_ms=4185.6, source="fsp-anns-parent/logs/New Results:54-57")}
        for wname in ("sift1m_P4_FAST", "sift1m_P10_HIGH"):
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", wname, "--k", "100", "--data", args.shipped_data, "--steps", str(args.shipped_steps),
                   "--warmup", "3", "--prewarm", "10", "--cpu-sample", "64", "--no-shipped", "--solo-tail", "0"]
            t_c = time.perf_counter()
            try:
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
                line = [x for x in pr.stdout.splitlines() if x.startswith("{")]
                if pr.returncode != 0 or not line:
                    shipped_out[wname] = dict(error=("rc %d: " % pr.returncode) + pr.stderr[-400:])
                    continue
                cj = json.loads(line[-1])
                st_ = cj.get("stages_ms") or {}
                tot_ = sum(v for v in st_.values() if v) or 1.0
                rf_ = cj.get("roofline") or {}
                cb_ = cj.get("cpu_baseline") or {}
                osf_ = (cj.get("operator_surface") or {})
                shipped_out[wname] = dict(
                    value=cj["value"], unit=cj["unit"], ms_per_step=cj["ms_per_step"], steps=cj["steps"], data=cj["data"],
                    config={k_: cj["config"][k_] for k_ in ("tables", "divisions", "m", "lambda", "probes", "hard_cap", "B", "k", "queries_per_step", "pipeline")},
                    recall_at_k=cj.get("recall_at_k"),
                    reference_published=ref_pub[wname],
                    scan=dict(kernel=rf_.get("kernel"), frac=rf_.get("frac"), achieved=rf_.get("achieved"), avg_launch_ms=rf_.get("avg_launch_ms"),
                              launches=rf_.get("launches"), algorithmic_bytes_per_launch=rf_.get("algorithmic_bytes_per_launch")),
                    stages_ms=st_, route_share_of_serial_step=round((st_.get("route_select") or 0.0) / tot_, 3),
                    cpu_baseline=dict(value=cb_.get("value"), unit=cb_.get("unit"), cores=cb_.get("cores"), kind=cb_.get("kind"), sample=cb_.get("sample"),
                                      matches_gpu=cb_.get("matches_gpu")),
                    operator_surface_per_query_us=(osf_.get("per_query") or {}).get("route_plus_refine"),
                    treeified=cj.get("treeified"), wall_s=round(time.perf_counter() - t_c, 1))
            except Exception as e:      # a report beside the line, never a reason to lose it
                shipped_out[wname] = dict(error=str(e)[:300])
        # ... and the HEADLINE configuration itself on the SIFT-like data (north_star quotes throughput on random Gaussian vectors, where no
        # LSH has signal: this line says what the same kernels do when the tables agree on some candidates, and what recall B = 256 buys)
        other = {}
        if args.data == "gaussian" and args.workload == "sift1m_T16_b32_B256_Q1024":
            cmd = [sys.executable, os.path.abspath(__file__), "--data", args.shipped_data, "--steps", "200", "--warmup", "10", "--no-shipped", "--no-extras",
                   "--no-cpu-baseline"]
            try:
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                line = [x for x in pr.stdout.splitlines() if x.startswith("{")]
                if pr.returncode == 0 and line:
                    cj = json.loads(line[-1])
                    other = dict(value=cj["value"], unit=cj["unit"], ms_per_step=cj["ms_per_step"], steps=cj["steps"], data=cj["data"],
                                 recall_at_10=cj.get("recall_at_10"), distance_ratio_at_10=cj.get("distance_ratio_at_10"), stages_ms=cj.get("stages_ms"),
                                 scan_frac=(cj.get("roofline") or {}).get("frac"), route=(cj.get("route_stage") or {}).get("kernels"))
                else:
                    other = dict(error=("rc %d: " % pr.returncode) + pr.stderr[-400:])
            except Exception as e:
                other = dict(error=str(e)[:300])
        extra = dict(shipped_profiles=shipped_out, headline_config_on_siftlike_data=other,
                     note="the reference's shipped SIFT1M profiles run by this same command as child processes (full select + chunked scan + "
                          "merge, three kernels per step on three contexts); `value` above is untouched by them")

    if rank == 0:
        out = {
            "metric": (("queries/sec @ recall@%d, SIFT-1M-shaped synthetic data d=%d B=%d" % (k, d, B)) if (args.data in ("gaussian", "clustered") or args.data.startswith("siftlike"))
                       else "queries/sec @ recall@%d, d=%d B=%d" % (k, d, B)),
            "value": round(qps, 1),
            "unit": "queries/s",
            "all_ranks_verified": ranks_verified,   # N > 1 (or --gpus 1 under torch.distributed.run): every rank compared every slice of the gathered top-k
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "host_issue_ms_per_step": round(host_issue_main * 1000.0 / args.steps, 4) if host_issue_main else None,   # launch issue alone (Python + HIP runtime): if it is close to ms_per_step the region is host-bound
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": data_desc + ", exact-kNN ground truth of the same set",
            "config": {"workload": args.workload, "n": n, "dim": d, "tables": T, "divisions": D, "m": m, "lambda": lam,
                       "code_bits": m * lam, "probes": P_eff, "hard_cap": HARD_CAP, "B": B, "k": k, "queries_per_gpu_per_step": Q, "queries_per_step": q_job,
                       "distinct_query_batches": NB, "route_counters": bool(args.route_counters), "passes_per_step": len(probe_passes),
                       "parallelism": f"query-sharded x{world}, index replicated", "merge": gather_path, "streams_per_gpu": nctx,
                       "pipeline": ("tick: 3 batches in flight, one launch per step = encode(t+2) + Route(t+1) + Refine(t) as one kernel (tick_kernel, "
                                    "fused=%s); every step does one full-batch encode, Route and Refine" % tick_fused) if use_tick
                       else ("front: %d contexts per GPU take the batches in turn (consecutive steps = different batches), each running encode(its next "
                             "batch) + Route(this batch) as ONE launch (front_kernel: encode workgroups beside the bounded select's) and then the scan "
                             "of this batch (whose workgroups finish queries the bounded select handed over: no hand-back launch), on its own HIP "
                             "stream" % nctx) if args.pipeline == "front"
                       else ("concurrent: %d contexts per GPU take the batches in turn, each running encode -> Route -> Refine of its batch as three "
                             "kernels on its own HIP stream (hardware queues overlap Route of one batch with Refine of another)" % nctx) if nctx > 1
                       else "serial: encode, Route, Refine of one batch as three kernels on one stream",
                       "candidates": {"dense": "kernel path (SURVEY 8d): [Q][B][d] blocks of decrypted candidate rows resident in HBM before the timed "
                                               "region (packed once per distinct batch), scanned by refine_stream_kernel",
                                      "store": "trusted-HBM variant: rows read from an HBM-resident plaintext store by id inside the refine scan",
                                      "gather": "rows packed into [Q][B][d] by a gather kernel inside the step, then scanned"}[mode]},
            "recall_at_10": recall if k == 10 else None,
            "distance_ratio_at_10": ratio if k == 10 else None,
            "recall_at_k": {"k": k, "recall": recall, "distance_ratio": ratio},
            "treeified": {"finished_on_host_while_packing": int(treeified_setup), "finished_on_host_in_checked_batch": int(treeified_batch0),
                          "flagged_and_left_empty_in_all_timed_and_untimed_runs": int(flagged_in_timed_runs),
                          "note": "queries whose HashMap<String,Long> bestScore treeifies a bin: flagged by the full select (count -1), finished by "
                                  "the library's JDK model on the host (fspann_route_resolve_dev), outside the timed region"},
            "recall_sweep": recall_sweep,
            "setup": {"build_index_s": round(setup_build_s, 3),
                      "note": "fspann_build_index of the whole base set: H2D of the vectors, coding (MFMA pre-filter + exact re-check), "
                              "radix sorts + partition cut of every table on the GPU, treeify replay of the staging map on the host"},
            "groundtruth": {"kernel": "gt_dist_kernel + gt_select_kernel (exact, GroundtruthPrecompute semantics)", "queries": Q, "base": n,
                            "ms": round(gt_ms, 1) if gt_ms is not None else None},
            "stages_ms": {"encode": round(float(st_mean[0]), 5), "route_select": round(float(st_mean[1]), 5),
                          "stage_candidates": round(float(st_mean[2]), 5), "refine_topk": round(float(st_mean[3]), 5)},
            "roofline": roofline,
            "encode_stage": encode_stage,
            "route_stage": route_info,
            "variants": variants or None,
            "end_to_end": end_to_end,
            "operator_surface": operator_surface,
            "timed_check": timed_check,
            "cpu_baseline": cpu,
            "extra": extra,
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    for cm in (comms or []):
        cm.close()
    for c_ in ctxs[::-1]:
        c_.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
