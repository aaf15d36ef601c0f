"""ctypes binding for the CPU oracle (oracle/fspann_oracle.cpp).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by the product package.
Parity status: "parity unpinned" (see the header of fspann_oracle.cpp).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "fspann_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_splitmix_next.restype = C.c_uint64
        L.orc_splitmix_next_double.restype = C.c_double
        L.orc_d2i.restype = C.c_int32
        L.orc_d2i.argtypes = [C.c_double]
        L.orc_string_hash.restype = C.c_int32
        L.orc_string_hash.argtypes = [C.c_char_p]
        L.orc_decimal_hash.restype = C.c_int32
        L.orc_decimal_hash.argtypes = [C.c_int64]
        L.orc_table_size_for.restype = C.c_int32
        L.orc_table_size_for.argtypes = [C.c_int32]
        L.orc_compute_key.restype = C.c_int64
        L.orc_hamming.restype = C.c_int64
        L.orc_pq_trace.restype = C.c_int64
        L.orc_ctx_create.restype = C.c_void_p
        L.orc_index_nparts.restype = C.c_int64
        L.orc_index_nids.restype = C.c_int64
        L.orc_route.restype = C.c_int64
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ---- Java-semantics hooks --------------------------------------------------
def splitmix_stream(seed: int, n: int):
    st = C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF)
    return [lib().orc_splitmix_next(C.byref(st)) for _ in range(n)]


def splitmix_doubles(seed: int, n: int):
    st = C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF)
    return [lib().orc_splitmix_next_double(C.byref(st)) for _ in range(n)]


def d2i(x: float) -> int:
    return lib().orc_d2i(x)


def string_hash(s: str) -> int:
    return lib().orc_string_hash(s.encode("ascii"))


def decimal_hashes(n: int) -> np.ndarray:
    out = np.empty(n, np.int32)
    lib().orc_decimal_hashes(C.c_int64(n), _p(out))
    return out


def table_size_for(c: int) -> int:
    return lib().orc_table_size_for(c)


def hashmap_order_ex(initial_capacity: int, keys, hashes, decimal=False):
    """Iteration order of `keys` put in that order into new HashMap<>(initial_capacity) — tree bins included.
    Returns (order, final table length, treeified, unmodelled): treeified = some bin became a red-black tree (modelled);
    unmodelled = a tree bin needed String.compareTo between two different keys with equal hashCode and the keys are not
    decimal ordinals (decimal=True: the key IS the number whose decimal string is the id)."""
    keys = _c(keys, np.int32)
    hashes = _c(hashes, np.int32)
    out = np.empty_like(keys)
    cap = C.c_int32(0)
    fl = lib().orc_hashmap_order(C.c_int32(initial_capacity), C.c_int64(len(keys)), _p(keys), _p(hashes),
                                 _p(out), C.byref(cap), C.c_int(1 if decimal else 0))
    return out, cap.value, bool(fl & 1), bool(fl & 2)


def hashmap_order(initial_capacity: int, keys, hashes, decimal=False):
    """(order, final table length, treeified) — see hashmap_order_ex."""
    out, cap, tree, _ = hashmap_order_ex(initial_capacity, keys, hashes, decimal)
    return out, cap, tree


def pq_trace(ops):
    ops = _c(ops, np.int64)
    out = np.empty(len(ops), np.int64)
    k = lib().orc_pq_trace(C.c_int64(len(ops)), _p(ops), _p(out))
    return out[:k].copy()


def compute_key(words) -> int:
    w = _c(words, np.uint64)
    return lib().orc_compute_key(_p(w), C.c_int(len(w)))


def hamming(a, b) -> int:
    a = _c(a, np.uint64)
    b = _c(b, np.uint64)
    return lib().orc_hamming(_p(a), _p(b), C.c_int(len(a)))


def quickcheck():
    h0 = C.c_int32(0)
    b0 = C.c_int(0)
    rc = lib().orc_quickcheck(C.byref(h0), C.byref(b0))
    return rc, h0.value, b0.value


# ---- GFunctions --------------------------------------------------------------
def build_random_g(d, m, omega, seed):
    alpha = np.empty((m, d), np.float64)
    r = np.empty(m, np.float64)
    w = np.empty(m, np.float64)
    lib().orc_build_random_g(C.c_int(d), C.c_int(m), C.c_double(omega), C.c_int64(seed), _p(alpha), _p(r), _p(w))
    return alpha, r, w


def build_from_sample(sample, m, seed):
    sample = _c(sample, np.float64)
    ns, d = sample.shape
    alpha = np.empty((m, d), np.float64)
    r = np.empty(m, np.float64)
    w = np.empty(m, np.float64)
    lib().orc_build_from_sample(_p(sample), C.c_int(ns), C.c_int(d), C.c_int(m), C.c_int64(seed), _p(alpha),
                                _p(r), _p(w))
    return alpha, r, w


def registry_init(sample, m, base_seed, T, D):
    """GFunctionRegistry.initialize: returns alpha[T*D,m,d], r[T*D,m], omega[T*D,m]."""
    sample = _c(sample, np.float64)
    ns, d = sample.shape
    alpha = np.empty((T * D, m, d), np.float64)
    r = np.empty((T * D, m), np.float64)
    w = np.empty((T * D, m), np.float64)
    lib().orc_registry_init(_p(sample), C.c_int(ns), C.c_int(d), C.c_int(m), C.c_int64(base_seed), C.c_int(T),
                            C.c_int(D), _p(alpha), _p(r), _p(w))
    return alpha, r, w


def H(v, alpha, r, omega):
    v = _c(v, np.float64)
    alpha = _c(alpha, np.float64)
    m, d = alpha.shape
    out = np.empty(m, np.int32)
    rc = lib().orc_H(_p(v), C.c_int(d), C.c_int(m), _p(alpha), _p(_c(r, np.float64)), _p(_c(omega, np.float64)),
                     _p(out))
    if rc != 0:
        raise ValueError("Vector contains NaN/Inf")
    return out


def Ccode(v, alpha, r, omega, lam):
    v = _c(v, np.float64)
    alpha = _c(alpha, np.float64)
    m, d = alpha.shape
    W = (m * lam + 63) // 64
    out = np.zeros(W, np.uint64)
    rc = lib().orc_C(_p(v), C.c_int(d), C.c_int(m), C.c_int(lam), _p(alpha), _p(_c(r, np.float64)),
                     _p(_c(omega, np.float64)), _p(out))
    if rc != 0:
        raise ValueError("Vector contains NaN/Inf")
    return out


class Oracle:
    """One reference 'system': registry + PartitionedIndexService + QueryServiceImpl, restated."""

    def __init__(self, T, D, m, lam, d, max_global_candidates=20000, refinement_limit=20000,
                 probe_override=-1, hamming_threshold=0):
        self.T, self.D, self.m, self.lam, self.d = T, D, m, lam, d
        self.TD = T * D
        self.W = (m * lam + 63) // 64
        self.hard_cap = max(max_global_candidates, refinement_limit)
        self.refinement_limit = refinement_limit
        self._h = C.c_void_p(lib().orc_ctx_create(T, D, m, lam, d, max_global_candidates, refinement_limit,
                                                  probe_override, hamming_threshold))
        self.n = 0

    def close(self):
        if self._h:
            lib().orc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def unmodelled(self) -> bool:
        """A tree bin needed the order of two different ids with equal hashCode whose Strings are unknown (non-decimal ids)."""
        return bool(lib().orc_unmodelled(self._h))

    @property
    def treeified(self) -> bool:
        """Some HashMap of this context turned a bin into a red-black tree (modelled)."""
        return bool(lib().orc_treeified(self._h))

    def set_gfunctions(self, alpha, r, omega):
        self.alpha = _c(alpha, np.float64).reshape(self.TD, self.m, self.d)
        self.r = _c(r, np.float64).reshape(self.TD, self.m)
        self.omega = _c(omega, np.float64).reshape(self.TD, self.m)
        lib().orc_set_gfunctions(self._h, _p(self.alpha), _p(self.r), _p(self.omega))

    def set_id_meta(self, n, java_hash=None, deleted=None):
        self.n = n
        jh = None if java_hash is None else _c(java_hash, np.int32)
        dl = None if deleted is None else _c(deleted, np.uint8)
        lib().orc_set_id_meta(self._h, C.c_int64(n), _p(jh), _p(dl))

    def set_store(self, vecs, valid=None):
        vecs = _c(vecs, np.float64)
        vl = None if valid is None else _c(valid, np.uint8)
        lib().orc_set_store(self._h, C.c_int64(vecs.shape[0]), _p(vecs), _p(vl))

    def encode(self, q):
        q = _c(q, np.float64).reshape(-1, self.d)
        codes = np.zeros((q.shape[0], self.TD, self.W), np.uint64)
        rc = lib().orc_encode(self._h, C.c_int64(q.shape[0]), _p(q), _p(codes))
        if rc != 0:
            raise ValueError("Vector contains NaN/Inf")
        return codes

    def hashes(self, q):
        q = _c(q, np.float64).reshape(-1, self.d)
        Hh = np.zeros((q.shape[0], self.TD, self.m), np.int32)
        lib().orc_hashes(self._h, C.c_int64(q.shape[0]), _p(q), _p(Hh))
        return Hh

    @staticmethod
    def staged_order(n, min_sample=1000):
        out = np.empty(n, np.int32)
        lib().orc_staged_order(C.c_int64(n), C.c_int64(min_sample), _p(out))
        return out

    def build_index(self, vectors, order=None):
        """PIS.insert ... finalizeForSearch.  `order` = staged order of handles."""
        vectors = _c(vectors, np.float64)
        n = vectors.shape[0]
        if order is None:
            order = self.staged_order(n)
        order = _c(order, np.int32)
        codes = self.encode(vectors[order])
        lib().orc_build_index(self._h, C.c_int64(len(order)), _p(order), _p(codes))
        return codes

    def get_index(self, td):
        npart = lib().orc_index_nparts(self._h, C.c_int(td))
        nid = lib().orc_index_nids(self._h, C.c_int(td))
        mn = np.empty(npart, np.int64)
        mx = np.empty(npart, np.int64)
        rep = np.empty((npart, self.W), np.uint64)
        off = np.empty(npart + 1, np.int64)
        ids = np.empty(nid, np.int32)
        lib().orc_get_index(self._h, C.c_int(td), _p(mn), _p(mx), _p(rep), _p(off), _p(ids))
        return dict(min_key=mn, max_key=mx, rep=rep, id_off=off, ids=ids)

    def set_index(self, td, min_key, max_key, rep, id_off, ids):
        mn = _c(min_key, np.int64)
        mx = _c(max_key, np.int64)
        rp = _c(rep, np.uint64)
        of = _c(id_off, np.int64)
        ii = _c(ids, np.int32)
        lib().orc_set_index(self._h, C.c_int(td), C.c_int64(len(mn)), _p(mn), _p(mx), _p(rp), _p(of), _p(ii))

    def route(self, codes, probe_override=-1, truncate=False, cap=None):
        codes = _c(codes, np.uint64).reshape(-1, self.TD, self.W)
        nq = codes.shape[0]
        if cap is None:
            cap = lib().orc_route(self._h, C.c_int64(nq), _p(codes), C.c_int(probe_override), C.c_int(int(truncate)),
                                  C.c_int64(0), None, None, None, None)
            cap = max(int(cap), 1)
        ids = np.full((nq, cap), -1, np.int32)
        score = np.full((nq, cap), -1, np.int32)
        count = np.zeros(nq, np.int32)
        raw = np.zeros(nq, np.int32)
        lib().orc_route(self._h, C.c_int64(nq), _p(codes), C.c_int(probe_override), C.c_int(int(truncate)),
                        C.c_int64(cap), _p(ids), _p(score), _p(count), _p(raw))
        return ids, score, count, raw

    def route_treeified(self, codes, probe_override=-1):
        """Per query: True where java.util.HashMap would have treeified a bin of bestScore (order unmodelled)."""
        codes = _c(codes, np.uint64).reshape(-1, self.TD, self.W)
        flags = np.zeros(codes.shape[0], np.uint8)
        lib().orc_route_treeified(self._h, C.c_int64(codes.shape[0]), _p(codes), C.c_int(probe_override), _p(flags))
        return flags.astype(bool)

    def search(self, q, K, codes=None, probe_override=-1, refine_override=0, sel_cap=None, threads=0):
        q = _c(q, np.float64).reshape(-1, self.d)
        nq = q.shape[0]
        if codes is None:
            codes = self.encode(q)
        codes = _c(codes, np.uint64)
        if sel_cap is None:
            sel_cap = refine_override if refine_override > 0 else self.refinement_limit
        out_ids = np.empty((nq, K), np.int32)
        out_dist = np.empty((nq, K), np.float64)
        out_count = np.empty(nq, np.int32)
        sel = np.full((nq, sel_cap), -1, np.int32)
        sel_count = np.zeros(nq, np.int32)
        metrics = np.zeros((nq, 5), np.int32)
        lib().orc_search(self._h, C.c_int64(nq), _p(q), _p(codes), C.c_int(K), C.c_int(probe_override),
                         C.c_int(refine_override), _p(out_ids), _p(out_dist), _p(out_count), _p(sel), _p(sel_count),
                         C.c_int64(sel_cap), _p(metrics), C.c_int(threads))
        return dict(ids=out_ids, dist=out_dist, count=out_count, sel=sel, sel_count=sel_count, metrics=metrics)


def refine(q, cand, cand_ids, cand_count, K):
    q = _c(q, np.float64)
    cand = _c(cand, np.float64)
    nq, B, d = cand.shape
    cand_ids = _c(cand_ids, np.int32)
    cand_count = _c(cand_count, np.int32)
    out_ids = np.empty((nq, K), np.int32)
    out_dist = np.empty((nq, K), np.float64)
    out_count = np.empty(nq, np.int32)
    lib().orc_refine(C.c_int64(nq), C.c_int(d), C.c_int64(B), _p(q), _p(cand), _p(cand_ids), _p(cand_count),
                     C.c_int(K), _p(out_ids), _p(out_dist), _p(out_count))
    return out_ids, out_dist, out_count


def num_threads() -> int:
    return lib().orc_num_threads()


def groundtruth(base, q, k):
    """GroundtruthPrecompute.run restated: ids [nq][k] (ties by lower id), squared distances."""
    base = _c(base, np.float32)
    q = _c(q, np.float32)
    n, d = base.shape
    ids = np.empty((q.shape[0], k), np.int32)
    d2 = np.empty((q.shape[0], k), np.float64)
    lib().orc_groundtruth(C.c_int64(n), _p(base), C.c_int64(q.shape[0]), _p(q), C.c_int(d), C.c_int(k), _p(ids), _p(d2))
    return ids, d2


def metrics(base, q, k, ann, ann_count, gt):
    """ForwardSecureANNSystem.computeMetricsAtK restated: (recall@k, distance ratio@k) per query."""
    base = _c(base, np.float32)
    q = _c(q, np.float32)
    ann = _c(ann, np.int32)
    gt = _c(gt, np.int32)
    cnt = None if ann_count is None else _c(ann_count, np.int32)
    rec = np.empty(q.shape[0], np.float64)
    rat = np.empty(q.shape[0], np.float64)
    lib().orc_metrics(C.c_int64(base.shape[0]), _p(base), C.c_int64(q.shape[0]), _p(q), C.c_int(base.shape[1]), C.c_int(k), _p(ann),
                      C.c_int64(ann.shape[1]), _p(cnt), _p(gt), C.c_int64(gt.shape[1]), _p(rec), _p(rat))
    return rec, rat
