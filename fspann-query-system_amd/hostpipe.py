"""Host candidate pipeline (include/fspann.h, "host candidate pipeline"): the packed point store with AES-256-GCM records
in the reference's exact format, and the three-stage Route | decrypt | Refine pipeline over one context.  Product code —
never imports oracle/.  Decrypt stays on the HOST (north_star): the GPU sees plaintext rows only as Refine's input block.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native as N


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class PointStore:
    """n records of dimension dim: {key version, IV, ciphertext || tag} per id handle (id string = decimal handle)."""

    def __init__(self, n: int, dim: int, master_key: bytes | None = None):
        self.L = N.lib()
        h = C.c_void_p()
        N.check(self.L.fspann_pointstore_create(n, dim, C.byref(h)))
        self._h, self.n, self.dim = h, n, dim
        self.set_master_key(master_key or os.urandom(32))

    def close(self):
        if getattr(self, "_h", None):
            self.L.fspann_pointstore_destroy(self._h)
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

    def set_master_key(self, key: bytes):
        assert len(key) == 32
        self.master = bytes(key)
        N.check(self.L.fspann_pointstore_set_master_key(self._h, C.c_char_p(self.master)))

    @property
    def version(self) -> int:
        return int(self.L.fspann_pointstore_current_version(self._h))

    def rotate(self) -> int:
        v = C.c_int(0)
        N.check(self.L.fspann_pointstore_rotate(self._h, C.byref(v)))
        return v.value

    def retire(self, version: int):
        N.check(self.L.fspann_pointstore_retire(self._h, version))

    def encrypt(self, vectors, h0: int = 0, threads: int = 0):
        v = np.ascontiguousarray(vectors)
        if v.dtype not in (np.float32, np.float64):
            v = v.astype(np.float64)
        v = v.reshape(-1, self.dim)
        N.check(self.L.fspann_pointstore_encrypt(self._h, h0, len(v), _p(v), N.F32 if v.dtype == np.float32 else N.F64,
                                                 threads or (os.cpu_count() or 1)))

    def delete(self, h: int):
        N.check(self.L.fspann_pointstore_delete(self._h, h))

    def reencrypt(self, handles, threads: int = 1) -> int:
        hs = np.ascontiguousarray(handles, dtype=np.int32)
        done = C.c_int64(0)
        N.check(self.L.fspann_pointstore_reencrypt(self._h, _p(hs), len(hs), threads, C.byref(done)))
        return int(done.value)

    def open_batch(self, ids, count, dtype=np.float32, threads: int = 0):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        nq, B = ids.shape
        count = np.ascontiguousarray(count, dtype=np.int32)
        dst = np.zeros((nq, B, self.dim), dtype)
        oi = np.full((nq, B), -1, np.int32)
        oc = np.zeros(nq, np.int32)
        N.check(self.L.fspann_pointstore_open_batch(self._h, nq, B, _p(ids), _p(count), _p(dst), N.F32 if dtype == np.float32 else N.F64,
                                                    _p(oi), _p(oc), threads or (os.cpu_count() or 1)))
        return dst, oi, oc

    def get_record(self, h: int):
        ver = C.c_int32(0)
        iv = np.zeros(12, np.uint8)
        ct = np.zeros(8 * self.dim + 16, np.uint8)
        N.check(self.L.fspann_pointstore_get_record(self._h, h, C.byref(ver), _p(iv), _p(ct)))
        return ver.value, iv.tobytes(), ct.tobytes()

    def put_record(self, h: int, version: int, iv: bytes, ct: bytes):
        assert len(iv) == 12 and len(ct) == 8 * self.dim + 16
        N.check(self.L.fspann_pointstore_put_record(self._h, h, version, C.c_char_p(iv), C.c_char_p(ct)))

    def stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        N.check(self.L.fspann_pointstore_stats(self._h, C.byref(a), C.byref(b)))
        return dict(opened=int(a.value), failed=int(b.value))


class Pipeline:
    """Route of batch i+1 | AES-GCM open of batch i on `host_threads` threads | H2D + Refine of batch i-1."""

    def __init__(self, ctx, store: PointStore, nq_max: int, B: int, k: int, host_threads: int = 0):
        self.L = N.lib()
        self.ctx, self.store, self.nq_max, self.B, self.k = ctx, store, nq_max, B, k
        self.threads = host_threads or (os.cpu_count() or 1)
        h = C.c_void_p()
        N.check(self.L.fspann_pipeline_create(ctx.handle, store.handle, nq_max, B, k, self.threads, C.byref(h)))
        self._h = h
        self.in_flight = 0

    def close(self):
        if getattr(self, "_h", None):
            self.L.fspann_pipeline_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def submit(self, q) -> int:
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, self.ctx.cfg.dim)
        t = C.c_uint64(0)
        N.check(self.L.fspann_pipeline_submit(self._h, len(q), _p(q), C.byref(t)))
        self.in_flight += 1
        return int(t.value)

    def collect(self):
        t, nq = C.c_uint64(0), C.c_int64(0)
        ids = np.empty((self.nq_max, self.k), np.int32)
        dist = np.empty((self.nq_max, self.k), np.float64)
        cnt = np.empty(self.nq_max, np.int32)
        N.check(self.L.fspann_pipeline_collect(self._h, C.byref(t), C.byref(nq), _p(ids), _p(dist), _p(cnt)))
        self.in_flight -= 1
        n = int(nq.value)
        return dict(ticket=int(t.value), ids=ids[:n], dist=dist[:n], count=cnt[:n])

    def stats(self):
        a, b, c, n = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int64(0)
        N.check(self.L.fspann_pipeline_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return dict(route_ms=a.value, decrypt_ms=b.value, refine_ms=c.value, batches=int(n.value))
