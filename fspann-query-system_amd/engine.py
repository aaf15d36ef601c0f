"""FspannContext — numpy-facing wrapper of one `fspann_ctx` (one GPU, one HIP stream).

Batch-first: every call takes nq >= 1 queries.  Host arrays go through the
host-pointer entry points of include/fspann.h; `*_dev` methods take raw device
pointers (ints, e.g. torch.Tensor.data_ptr()) and only enqueue work on the
context's stream.  Product code — never imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _native as N


@dataclass
class PaperRuntimeConfig:
    """The nine knobs the path reads (config/SystemConfig.java:237-338)."""
    tables: int = 6
    divisions: int = 3
    m: int = 24
    lambda_: int = 2
    dim: int = 128
    seed: int = 13
    refinement_limit: int = 20000
    max_global_candidates: int = 20000
    probe_override: int = -1
    hamming_prefilter_threshold: int = 0
    block_size: int = 64
    default_probes: int = 5

    def to_c(self) -> N.Cfg:
        return N.Cfg(self.tables, self.divisions, self.m, self.lambda_, self.dim, self.block_size,
                     self.default_probes, self.probe_override, self.max_global_candidates,
                     self.refinement_limit, self.hamming_prefilter_threshold, 0)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dt(a):
    if a.dtype == np.float32:
        return N.F32
    if a.dtype == np.float64:
        return N.F64
    raise N.FspannArgumentError(f"unsupported dtype {a.dtype}")


class FspannContext:
    def __init__(self, cfg: PaperRuntimeConfig, device: int = 0):
        self.cfg = cfg
        self.L = N.lib()
        h = C.c_void_p()
        cc = cfg.to_c()
        N.check(self.L.fspann_ctx_create(device, C.byref(cc), C.byref(h)))
        self._h = h
        self.TD = cfg.tables * cfg.divisions
        self.bits = cfg.m * cfg.lambda_
        self.W = (self.bits + 63) // 64
        self.hard_cap = max(cfg.max_global_candidates, cfg.refinement_limit)
        self.device = device

    # -- lifecycle -----------------------------------------------------------
    def clone(self) -> "FspannContext":
        """A context on the same device that reads THIS context's GFunctions, frozen index, id metadata and store in place
        (fspann_ctx_clone) and owns its stream and work areas: one index in HBM served from several streams."""
        other = FspannContext.__new__(FspannContext)
        other.cfg, other.L = self.cfg, self.L
        h = C.c_void_p()
        N.check(self.L.fspann_ctx_clone(self._h, C.byref(h)))
        other._h = h
        other.TD, other.bits, other.W, other.hard_cap, other.device = self.TD, self.bits, self.W, self.hard_cap, self.device
        return other

    def close(self):
        if getattr(self, "_h", None):
            self.L.fspann_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    @property
    def stream(self) -> int:
        return int(self.L.fspann_ctx_stream(self._h) or 0)

    def sync(self):
        N.check(self.L.fspann_sync(self._h))

    # -- Setup ---------------------------------------------------------------
    def set_gfunctions(self, alpha, r, omega):
        c = self.cfg
        a = _c(alpha, np.float64).reshape(self.TD, c.m, c.dim)
        rr = _c(r, np.float64).reshape(self.TD, c.m)
        ww = _c(omega, np.float64).reshape(self.TD, c.m)
        N.check(self.L.fspann_set_gfunctions(self._h, _p(a), _p(rr), _p(ww)))

    def registry_initialize(self, sample, base_seed=None):
        s = _c(sample, np.float64).reshape(-1, self.cfg.dim)
        seed = self.cfg.seed if base_seed is None else base_seed
        N.check(self.L.fspann_registry_initialize(self._h, _p(s), s.shape[0], seed))

    def get_gfunctions(self):
        c = self.cfg
        a = np.empty((self.TD, c.m, c.dim), np.float64)
        r = np.empty((self.TD, c.m), np.float64)
        w = np.empty((self.TD, c.m), np.float64)
        N.check(self.L.fspann_get_gfunctions(self._h, _p(a), _p(r), _p(w)))
        return a, r, w

    def set_id_meta(self, n_ids, java_hash=None, deleted=None):
        jh = None if java_hash is None else _c(java_hash, np.int32)
        dl = None if deleted is None else _c(deleted, np.uint8)
        N.check(self.L.fspann_set_id_meta(self._h, n_ids, _p(jh), _p(dl)))
        self.n_ids = n_ids

    def set_index(self, td, min_key, max_key, rep, id_off, ids):
        mn, mx = _c(min_key, np.int64), _c(max_key, np.int64)
        rp, of, ii = _c(rep, np.uint64), _c(id_off, np.int64), _c(ids, np.int32)
        N.check(self.L.fspann_set_index(self._h, td, len(mn), _p(mn), _p(mx), _p(rp), _p(of), _p(ii)))

    def finalize(self):
        N.check(self.L.fspann_finalize(self._h))

    def build_index(self, vectors, order=None):
        v = np.ascontiguousarray(vectors)
        if v.dtype not in (np.float32, np.float64):
            v = v.astype(np.float64)
        v = v.reshape(-1, self.cfg.dim)
        o = None if order is None else _c(order, np.int32)
        N.check(self.L.fspann_build_index(self._h, v.shape[0], _p(v), _dt(v), _p(o)))

    def build_begin(self, n_total: int):
        """Incremental Setup: begin(n) -> append(rows of the next handles) ... -> finish(order) (fspann_build_begin / _append / _finish)."""
        N.check(self.L.fspann_build_begin(self._h, int(n_total)))

    def build_append(self, rows):
        v = np.ascontiguousarray(rows)
        if v.dtype not in (np.float32, np.float64):
            v = v.astype(np.float64)
        v = v.reshape(-1, self.cfg.dim)
        N.check(self.L.fspann_build_append(self._h, v.shape[0], _p(v), _dt(v)))

    def build_finish(self, order=None):
        o = None if order is None else _c(order, np.int32)
        N.check(self.L.fspann_build_finish(self._h, _p(o)))

    def set_deleted(self, handles, flag=True):
        """Live mirror of metadata.isDeleted (PIS:739): no un-freeze, works on the owner or any clone while they serve queries."""
        h = _c(handles, np.int32).reshape(-1)
        N.check(self.L.fspann_set_deleted(self._h, _p(h), len(h), 1 if flag else 0))

    def save_index(self, path: str):
        N.check(self.L.fspann_index_save(self._h, os.fsencode(path)))

    def load_index(self, path: str):
        N.check(self.L.fspann_index_load(self._h, os.fsencode(path)))

    def get_index(self, td):
        npart, nid = C.c_int64(), C.c_int64()
        N.check(self.L.fspann_index_dims(self._h, td, C.byref(npart), C.byref(nid)))
        mn = np.empty(npart.value, np.int64)
        mx = np.empty(npart.value, np.int64)
        rep = np.empty((npart.value, self.W), np.uint64)
        off = np.empty(npart.value + 1, np.int64)
        ids = np.empty(nid.value, np.int32)
        N.check(self.L.fspann_get_index(self._h, td, _p(mn), _p(mx), _p(rep), _p(off), _p(ids)))
        return dict(min_key=mn, max_key=mx, rep=rep, id_off=off, ids=ids)

    # -- TokenGen ----------------------------------------------------------------
    def encode(self, q, want_hashes=False):
        if q is None:
            raise N.FspannNullError("query vector is null")
        q = np.ascontiguousarray(q)
        if q.dtype not in (np.float32, np.float64):
            q = q.astype(np.float64)
        if q.size % self.cfg.dim != 0:
            raise N.FspannArgumentError(f"Expected vector length {self.cfg.dim}")
        q = q.reshape(-1, self.cfg.dim)
        nq = q.shape[0]
        codes = np.zeros((nq, self.TD, self.W), np.uint64)
        hs = np.zeros((nq, self.TD, self.cfg.m), np.int32) if want_hashes else None
        N.check(self.L.fspann_encode(self._h, nq, _p(q), _dt(q), _p(codes), _p(hs)))
        return (codes, hs) if want_hashes else codes

    def set_encode_mode(self, mode: int):
        """0 auto, 1 exact fp64, 2 MFMA fp32 + exact re-check (all bit-identical)."""
        N.check(self.L.fspann_set_encode_mode(self._h, mode))

    def last_encode_rechecked(self) -> int:
        return int(self.L.fspann_last_encode_rechecked(self._h))

    # -- Route ---------------------------------------------------------------------
    def effective_probes(self, probe_override=-1):
        return self.L.fspann_effective_probes(self._h, probe_override)

    def set_route_mode(self, mode):
        """0 auto, 1 full select only, 2 bounded select whenever legal (identical results)."""
        N.check(self.L.fspann_set_route_mode(self._h, int(mode)))

    def last_route_info(self):
        import ctypes as C
        lazy, ovf = C.c_int(0), C.c_int(0)
        N.check(self.L.fspann_last_route_info(self._h, C.byref(lazy), C.byref(ovf)))
        return dict(lazy=bool(lazy.value), overflowed=int(ovf.value))

    def unmodelled_queries(self, reset=True) -> int:
        """Queries flagged 'HashMap bin treeified' (count = -1) by Route calls since the last reset."""
        v = C.c_int64(0)
        N.check(self.L.fspann_unmodelled_queries(self._h, C.byref(v), 1 if reset else 0))
        return int(v.value)

    def route_max_candidates(self, probe_override=-1):
        return int(self.L.fspann_route_max_candidates(self._h, probe_override))

    def route(self, codes, probe_override=-1, limit=N.INT32_MAX, cap=None, counters=True, allow_unmodelled=False):
        """allow_unmodelled: return the per-query flags (count = -1: a HashMap bin would be treeified, the JVM's order
        is not modelled) instead of raising FspannStateError for the whole batch."""
        if codes is None:
            raise N.FspannStateError("MSANNP violation: QueryToken missing BitSet codes")
        codes = _c(codes, np.uint64).reshape(-1, self.TD, self.W)
        nq = codes.shape[0]
        if cap is None:
            cap = max(1, min(limit, self.route_max_candidates(probe_override)))
        ids = np.full((nq, cap), -1, np.int32)
        score = np.full((nq, cap), -1, np.int32)
        count = np.zeros(nq, np.int32)
        kept = np.zeros(nq, np.int32)
        raw = np.zeros(nq, np.int32)
        rc = self.L.fspann_route(self._h, nq, _p(codes), probe_override, min(limit, N.INT32_MAX), cap, _p(ids),
                                 _p(score), _p(count), _p(kept) if counters else None, _p(raw) if counters else None)
        if not (allow_unmodelled and rc == N.E_STATE and (count < 0).any()):   # outputs are complete in that case
            N.check(rc)
        if not counters:   # lastCandKept / rawSeen not requested: the bounded select may run
            return dict(ids=ids, score=score, count=count)
        return dict(ids=ids, score=score, count=count, kept=kept, raw_seen=raw)

    def route_flags(self, codes, probe_override=-1):
        """Diagnostics: which queries the full select FLAGS (count = -1: a HashMap bin of bestScore would be treeified) before the
        host model finishes them — fspann_route_dev with the counters requested (exact detection), nothing resolved."""
        codes = _c(codes, np.uint64).reshape(-1, self.TD, self.W)
        nq = codes.shape[0]
        cap = max(1, self.route_max_candidates(probe_override))
        bufs = []

        def dev(nbytes):
            p = C.c_void_p()
            N.check(self.L.fspann_dev_alloc(self._h, nbytes, C.byref(p)))
            bufs.append(p)
            return p
        try:
            d_codes, d_ids, d_cnt, d_kept = dev(codes.nbytes), dev(nq * cap * 4), dev(nq * 4), dev(nq * 4)
            N.check(self.L.fspann_h2d(self._h, d_codes, _p(codes), codes.nbytes))
            N.check(self.L.fspann_route_dev(self._h, nq, d_codes, probe_override, N.INT32_MAX, cap, d_ids, None, d_cnt, d_kept, None))
            count = np.zeros(nq, np.int32)
            N.check(self.L.fspann_d2h(self._h, _p(count), d_cnt, nq * 4))
            self.unmodelled_queries(reset=True)
        finally:
            for p in bufs:
                self.L.fspann_dev_free(self._h, p)
        return count < 0

    def route_flags_bounded(self, codes, limit, probe_override=-1):
        """Diagnostics: which queries end FLAGGED (count = -1) when fspann_route_dev runs the way stage A.5 calls it — first
        `limit` entries, no counters, so the bounded select runs where it is legal and hands over what it cannot hold or what
        its exact treeify check catches; the full select then flags a treeified bestScore map.  Nothing is resolved."""
        codes = _c(codes, np.uint64).reshape(-1, self.TD, self.W)
        nq = codes.shape[0]
        bufs = []

        def dev(nbytes):
            p = C.c_void_p()
            N.check(self.L.fspann_dev_alloc(self._h, nbytes, C.byref(p)))
            bufs.append(p)
            return p
        try:
            d_codes, d_ids, d_cnt = dev(codes.nbytes), dev(nq * limit * 4), dev(nq * 4)
            N.check(self.L.fspann_h2d(self._h, d_codes, _p(codes), codes.nbytes))
            N.check(self.L.fspann_route_dev(self._h, nq, d_codes, probe_override, limit, limit, d_ids, None, d_cnt, None, None))
            count = np.zeros(nq, np.int32)
            N.check(self.L.fspann_d2h(self._h, _p(count), d_cnt, nq * 4))
            self.unmodelled_queries(reset=True)
        finally:
            for p in bufs:
                self.L.fspann_dev_free(self._h, p)
        return count < 0

    # -- Refine ----------------------------------------------------------------------
    def refine(self, q, cand, cand_ids, cand_count, k):
        cand = np.ascontiguousarray(cand)
        if cand.dtype not in (np.float32, np.float64):
            cand = cand.astype(np.float64)
        nq, B, d = cand.shape
        q = _c(q, cand.dtype).reshape(nq, d)
        ci = _c(cand_ids, np.int32).reshape(nq, B)
        cc = _c(cand_count, np.int32).reshape(nq)
        out_ids = np.empty((nq, k), np.int32)
        out_dist = np.empty((nq, k), np.float64)
        out_count = np.empty(nq, np.int32)
        scored = np.empty(nq, np.int32)
        N.check(self.L.fspann_refine(self._h, nq, _p(q), _p(cand), _dt(cand), B, _p(ci), _p(cc), k, _p(out_ids),
                                     _p(out_dist), _p(out_count), _p(scored)))
        return dict(ids=out_ids, dist=out_dist, count=out_count, scored=scored)

    def host_buffer(self, shape, dtype=np.float64):
        """A numpy view of the context's PINNED host block (fspann_host_buffer), at least as large as `shape` x `dtype`: what the adapter
        packs decrypted candidate rows into (QSI:238-271) — rows handed to refine() from here travel by plain DMA.  The view dies with the
        next larger request or with the context."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.L.fspann_host_buffer(self._h, max(n, 16))
        if not p:
            raise MemoryError(self.L.fspann_last_error().decode())
        return np.frombuffer((C.c_char * n).from_address(p), dtype=dtype).reshape(shape)

    def refine_store(self, q, cand_ids, cand_count, k):
        """Refine with rows read from the resident store by id (no staging copy)."""
        ci = _c(cand_ids, np.int32)
        nq, B = ci.shape
        q = np.ascontiguousarray(q)
        if q.dtype not in (np.float32, np.float64):
            q = q.astype(np.float64)
        q = q.reshape(nq, self.cfg.dim)
        cc = _c(cand_count, np.int32).reshape(nq)
        out_ids = np.empty((nq, k), np.int32)
        out_dist = np.empty((nq, k), np.float64)
        out_count = np.empty(nq, np.int32)
        scored = np.empty(nq, np.int32)
        N.check(self.L.fspann_refine_store(self._h, nq, _p(q), _dt(q), B, _p(ci), _p(cc), k, _p(out_ids),
                                           _p(out_dist), _p(out_count), _p(scored)))
        return dict(ids=out_ids, dist=out_dist, count=out_count, scored=scored)

    # -- plaintext store (test / bench harness) --------------------------------------
    def store_set(self, vectors):
        v = np.ascontiguousarray(vectors)
        if v.dtype not in (np.float32, np.float64):
            v = v.astype(np.float64)
        v = v.reshape(-1, self.cfg.dim)
        N.check(self.L.fspann_store_set(self._h, v.shape[0], _p(v), _dt(v)))
        self.store_dtype = v.dtype

    def store_attach_dev(self, n, ptr, dtype):
        """Use caller-owned device rows [n][dim] as the store (no copy; keep them alive)."""
        N.check(self.L.fspann_store_attach_dev(self._h, int(n), ptr, dtype))

    def hbm_read_peak(self, nbytes=1 << 32, reps=5) -> float:
        """GB/s of a pure-load kernel over `nbytes` of HBM on this device (bench.py roofline.peak_measured)."""
        v = C.c_double(0.0)
        N.check(self.L.fspann_hbm_read_peak(self._h, int(nbytes), int(reps), C.byref(v)))
        return float(v.value)

    def hbm_read_window(self, nbytes=1 << 32, window=134823936, reps=20) -> float:
        """GB/s of the same pure-load kernel when one launch reads only `window` bytes of cold HBM (average over `reps`)."""
        v = C.c_double(0.0)
        N.check(self.L.fspann_hbm_read_window(self._h, int(nbytes), int(window), int(reps), C.byref(v)))
        return float(v.value)

    # -- device-pointer entry points (ints) ----------------------------------------------
    def encode_dev(self, nq, q_ptr, dtype, codes_ptr, hashes_ptr=0, bad_ptr=0):
        N.check(self.L.fspann_encode_dev(self._h, nq, q_ptr, dtype, codes_ptr, hashes_ptr or None, bad_ptr or None))

    def route_dev(self, nq, codes_ptr, probe_override, limit, cap, ids_ptr, score_ptr, count_ptr, kept_ptr, raw_ptr):
        N.check(self.L.fspann_route_dev(self._h, nq, codes_ptr, probe_override, limit, cap, ids_ptr, score_ptr or None,
                                        count_ptr, kept_ptr or None, raw_ptr or None))

    def route_resolve_dev(self, nq, codes_ptr, probe_override, limit, cap, ids_ptr, score_ptr, count_ptr, kept_ptr=0, raw_ptr=0) -> int:
        """Finish the queries an asynchronous Route call flagged (count -1: a HashMap bin treeified) with the host model; returns how many."""
        done = C.c_int64(0)
        N.check(self.L.fspann_route_resolve_dev(self._h, nq, codes_ptr, probe_override, limit, cap, ids_ptr, score_ptr or None, count_ptr,
                                                kept_ptr or None, raw_ptr or None, C.byref(done)))
        return int(done.value)

    def search_store_finish_dev(self, nq, q_ptr, q_dtype, probe_override, B, k, out_ids_ptr, out_dist_ptr, out_count_ptr, scored_ptr=0,
                                sel_ids_ptr=0, sel_count_ptr=0) -> int:
        """Completes the preceding search_store_dev call: flagged queries are finished on the host and the batch is scored again."""
        done = C.c_int64(0)
        N.check(self.L.fspann_search_store_finish_dev(self._h, nq, q_ptr, q_dtype, probe_override, B, k, out_ids_ptr, out_dist_ptr, out_count_ptr,
                                                      scored_ptr or None, sel_ids_ptr or None, sel_count_ptr or None, C.byref(done)))
        return int(done.value)

    def store_gather_dev(self, nq, sel_ids_ptr, sel_count_ptr, B, cand_ptr):
        N.check(self.L.fspann_store_gather_dev(self._h, nq, sel_ids_ptr, sel_count_ptr, B, cand_ptr))

    def search_store_dev(self, nq, q_ptr, q_dtype, probe_override, B, k, out_ids_ptr, out_dist_ptr, out_count_ptr, scored_ptr=0,
                         sel_ids_ptr=0, sel_count_ptr=0, bad_ptr=0):
        """encode -> route(limit = B) -> refine from the resident store, one call (device pointers, stream order)."""
        N.check(self.L.fspann_search_store_dev(self._h, nq, q_ptr, q_dtype, probe_override, B, k, out_ids_ptr, out_dist_ptr,
                                               out_count_ptr, scored_ptr or None, sel_ids_ptr or None, sel_count_ptr or None,
                                               bad_ptr or None))

    def groundtruth_dev(self, n, base_ptr, nq, q_ptr, dim, k, out_ids_ptr, out_d2_ptr=0):
        """Exact k-NN (GroundtruthPrecompute semantics) of device-resident fp32 base / query rows."""
        N.check(self.L.fspann_groundtruth_dev(self._h, n, base_ptr, nq, q_ptr, dim, k, out_ids_ptr, out_d2_ptr or None))

    def eval_metrics_dev(self, n, base_ptr, nq, q_ptr, dim, k, ann_ptr, ann_stride, ann_count_ptr, gt_ptr, gt_stride, recall_ptr, ratio_ptr):
        """recall@k / distance ratio@k per query (ForwardSecureANNSystem.computeMetricsAtK)."""
        N.check(self.L.fspann_eval_metrics_dev(self._h, n, base_ptr, nq, q_ptr, dim, k, ann_ptr, ann_stride, ann_count_ptr or None, gt_ptr, gt_stride,
                                               recall_ptr, ratio_ptr))

    def route_handover_bytes(self, nq, probe_override=-1) -> int:
        return int(self.L.fspann_route_handover_bytes(self._h, nq, probe_override))

    def tick_dev(self, encode=None, route=None, refine=None):
        """One launch for encode / Route / Refine of three batches in flight (fspann_tick_dev).  Each part is a dict of
        device pointers (ints) or None:
          encode: nq, q, codes, [dtype = F32], [bad]
          route:  nq, codes, limit, ids, count, [probe_override], [handover]
          refine: nq, q, B, ids, count, k, out_ids, out_dist, out_count, [cand (None: from the store)], [q_dtype], [cand_dtype],
                  [codes + handover of that batch], [probe_override], [scored]"""
        t = N.Tick()
        if encode:
            t.nq_encode, t.enc_q_dev, t.enc_dtype = encode["nq"], encode["q"], encode.get("dtype", N.F32)
            t.enc_codes_dev, t.enc_bad_dev = encode["codes"], encode.get("bad") or None
        if route:
            t.nq_route, t.route_codes_dev, t.route_limit = route["nq"], route["codes"], route["limit"]
            t.route_probe_override = route.get("probe_override", -1)
            t.route_ids_dev, t.route_count_dev, t.route_handover_dev = route["ids"], route["count"], route.get("handover") or None
        if refine:
            t.nq_refine, t.ref_q_dev, t.ref_B, t.k = refine["nq"], refine["q"], refine["B"], refine["k"]
            t.ref_q_dtype, t.ref_cand_dtype = refine.get("q_dtype", N.F32), refine.get("cand_dtype", N.F32)
            t.ref_cand_dev = refine.get("cand") or None
            t.ref_ids_dev, t.ref_count_dev = refine["ids"], refine["count"]
            t.ref_codes_dev, t.ref_handover_dev = refine.get("codes") or None, refine.get("handover") or None
            t.ref_probe_override = refine.get("probe_override", -1)
            t.out_ids_dev, t.out_dist_dev, t.out_count_dev = refine["out_ids"], refine["out_dist"], refine["out_count"]
            t.scored_dev = refine.get("scored") or None
        N.check(self.L.fspann_tick_dev(self._h, C.byref(t)))

    def last_tick_fused(self) -> bool:
        return bool(self.L.fspann_last_tick_fused(self._h))

    def refine_timing_begin(self, max_launches, every=1):
        N.check(self.L.fspann_refine_timing_begin(self._h, int(max_launches), int(every)))

    def refine_timing_end(self):
        """(dispatches, total ms) of the refinement-scan kernels launched since refine_timing_begin (kernel-attached HIP events)."""
        import ctypes as C
        n, ms = C.c_int(0), C.c_double(0.0)
        N.check(self.L.fspann_refine_timing_end(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def refine_store_dev(self, nq, q_ptr, q_dtype, B, cand_ids_ptr, cand_count_ptr, k, out_ids_ptr, out_dist_ptr,
                         out_count_ptr, scored_ptr=0):
        N.check(self.L.fspann_refine_store_dev(self._h, nq, q_ptr, q_dtype, B, cand_ids_ptr, cand_count_ptr, k,
                                               out_ids_ptr, out_dist_ptr, out_count_ptr, scored_ptr or None))

    def refine_dev(self, nq, q_ptr, q_dtype, cand_ptr, cand_dtype, B, cand_ids_ptr, cand_count_ptr, k, out_ids_ptr,
                   out_dist_ptr, out_count_ptr, scored_ptr=0):
        N.check(self.L.fspann_refine_dev(self._h, nq, q_ptr, q_dtype, cand_ptr, cand_dtype, B, cand_ids_ptr,
                                         cand_count_ptr, k, out_ids_ptr, out_dist_ptr, out_count_ptr,
                                         scored_ptr or None))
