"""Host-side mirror of the reference's TokenGen / Route / Refine operator surface.

Same class and method names, argument meaning and error behaviour as the Java
operators, with the arithmetic delegated to libfspann_hip.so:

  QueryTokenFactory.create/derive        qry/core/QueryTokenFactory.java:63,182
  PartitionedIndexService.*              idx/PartitionedIndexService.java:265-347,459-896
  QueryServiceImpl.search + getLast*     qry/service/QueryServiceImpl.java:101-352,417-474
  GFunctionRegistry (static)             idx/GFunctionRegistry.java:63-252

Java exception -> Python: IllegalStateException -> FspannStateError,
IllegalArgumentException -> FspannArgumentError (a ValueError),
NullPointerException -> FspannNullError (a TypeError).

AES-GCM, key versions and point storage stay on the host behind the same three
collaborators the reference wires in (CryptoService, KeyLifeCycleService,
RocksDBMetadataManager); `InMemoryHost` is the plaintext test double for them.
The JVM binding for the same C ABI is jni/fspann_jni.cpp;
this module is its Python twin, used by tests/ and bench.py.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, NamedTuple, Optional, Sequence

import numpy as np

from . import _native as N
from .engine import FspannContext, PaperRuntimeConfig

MIN_SAMPLE_SIZE = 1000   # PIS:50
MAX_SAMPLE_SIZE = 10000  # PIS:51


# --------------------------------------------------------------------------------------
# value types (common/QueryToken.java, QueryResult.java, EncryptedPoint.java)
# --------------------------------------------------------------------------------------
class QueryResult(NamedTuple):
    id: str
    distance: float


class CandidateWithScore(NamedTuple):  # PIS:82-89
    id: str
    hammingDist: int


@dataclass
class EncryptedPoint:
    id: str
    version: int
    iv: bytes
    ciphertext: bytes
    dim: int


@dataclass
class KeyVersion:
    version: int
    key: bytes


class QueryToken:
    """common/QueryToken.java:49-71 — bitCodes is uint64[T][D][W] (BitSet words)."""

    def __init__(self, bitCodes, iv, encryptedQuery, topK, numTables, dimension, version, lambda_, encryptionContext):
        self._bitCodes = None if bitCodes is None else np.array(bitCodes, dtype=np.uint64, copy=True)
        if iv is None or encryptedQuery is None:
            raise N.FspannNullError("iv/encryptedQuery")
        self._iv = bytes(iv)
        self._ct = bytes(encryptedQuery)
        self._topK = max(1, int(topK))
        self._numTables = max(1, int(numTables))
        self._dimension = int(dimension)
        self._version = int(version)
        self._lambda = int(lambda_)
        self._ctx = encryptionContext

    def getBitCodes(self):
        return None if self._bitCodes is None else self._bitCodes.copy()

    def setBitCodes(self, bc):
        self._bitCodes = None if bc is None else np.array(bc, dtype=np.uint64, copy=True)

    def getIv(self): return self._iv
    def getEncryptedQuery(self): return self._ct
    def getTopK(self): return self._topK
    def getNumTables(self): return self._numTables
    def getDimension(self): return self._dimension
    def getVersion(self): return self._version
    def getLambda(self): return self._lambda
    def getEncryptionContext(self): return self._ctx


# --------------------------------------------------------------------------------------
# config (the fields of config/SystemConfig.java the path reads)
# --------------------------------------------------------------------------------------
@dataclass
class SystemConfig:
    m: int = 24
    lambda_: int = 2
    divisions: int = 3
    tables: int = 6
    seed: int = 13
    refinementLimit: int = 20000
    maxGlobalCandidates: int = 20000
    probeOverride: int = -1
    hammingPrefilterThreshold: int = 0
    kVariants: Sequence[int] = (1, 10, 20, 40, 60, 80, 100)

    def getMaxK(self):
        return max(self.kVariants)

    def native(self, dim: int) -> PaperRuntimeConfig:
        return PaperRuntimeConfig(tables=self.tables, divisions=self.divisions, m=self.m, lambda_=self.lambda_,
                                  dim=dim, seed=self.seed, refinement_limit=self.refinementLimit,
                                  max_global_candidates=self.maxGlobalCandidates, probe_override=self.probeOverride,
                                  hamming_prefilter_threshold=self.hammingPrefilterThreshold)


# --------------------------------------------------------------------------------------
# host collaborators (unchanged subsystems) + plaintext test double
# --------------------------------------------------------------------------------------
class InMemoryHost:
    """Stand-in for AesGcmCryptoService + KeyRotationServiceImpl + RocksDBMetadataManager.

    Payload layout follows crypto/AesGcmCryptoService.java:240-277 (8*dim bytes, big-endian fp64)
    but is NOT encrypted: the host crypto is out of scope (north_star) and unchanged."""

    def __init__(self):
        self.points: Dict[str, EncryptedPoint] = {}
        self.deleted = set()
        self.version = 1
        self.load_failures = set()   # ids whose load/decrypt raises -> skipped (QSI:265-270)

    # KeyLifeCycleService
    def getCurrentVersion(self): return KeyVersion(self.version, b"\0" * 32)
    def getVersion(self, v): return KeyVersion(v, b"\0" * 32)

    # CryptoService
    @staticmethod
    def _enc(vec): return np.asarray(vec, dtype=">f8").tobytes()
    @staticmethod
    def _dec(b): return np.frombuffer(b, dtype=">f8").astype(np.float64)
    def encrypt(self, id, vector, kv=None):
        return EncryptedPoint(id, (kv or self.getCurrentVersion()).version, os.urandom(12), self._enc(vector), len(vector))
    def decryptFromPoint(self, ep, key):
        if ep.id in self.load_failures:
            raise RuntimeError("decrypt failed")
        return self._dec(ep.ciphertext)
    def encryptQuery(self, vec, key, iv): return self._enc(vec)
    def decryptQuery(self, ct, iv, key): return self._dec(ct)

    # RocksDBMetadataManager
    def saveEncryptedPoint(self, ep): self.points[ep.id] = ep
    def loadEncryptedPoint(self, id): return self.points.get(id)
    def isDeleted(self, id): return id in self.deleted
    def markDeleted(self, id): self.deleted.add(id)


# --------------------------------------------------------------------------------------
# GFunctionRegistry — process-wide static, like the reference
# --------------------------------------------------------------------------------------
class GFunctionRegistry:
    _init = False
    DIM = M = LAMBDA = TABLES = DIVISIONS = -1
    BASE_SEED = -1
    alpha = r = omega = None

    @classmethod
    def initialize(cls, sample, dimension, m, lambda_, baseSeed, tables, divisions, ctx: FspannContext = None):
        """idx/GFunctionRegistry.java:63-147.  `ctx` runs the projection pass on the GPU."""
        if sample is None:
            raise N.FspannNullError("sample")
        if len(sample) == 0:
            raise N.FspannArgumentError("Sample vectors cannot be empty")
        s = np.asarray(sample, dtype=np.float64)
        if s.ndim != 2 or s.shape[1] != dimension:
            raise N.FspannArgumentError(f"Mixed dimensions in GFunctionRegistry sample: expected {dimension}")
        if (cls._init and cls.DIM == dimension and cls.M == m and cls.LAMBDA == lambda_ and cls.BASE_SEED == baseSeed
                and cls.TABLES == tables and cls.DIVISIONS == divisions):
            return  # :86-95 same configuration -> no-op
        own = ctx is None
        if own:
            ctx = FspannContext(PaperRuntimeConfig(tables=tables, divisions=divisions, m=m, lambda_=lambda_, dim=dimension,
                                                   seed=baseSeed))
        try:
            ctx.registry_initialize(s, baseSeed)
            a, r, w = ctx.get_gfunctions()
        finally:
            if own:
                ctx.close()
        cls.install(a, r, w, dimension, m, lambda_, baseSeed, tables, divisions)

    @classmethod
    def install(cls, alpha, r, omega, dimension, m, lambda_, baseSeed, tables, divisions):
        """Import GFunctions generated elsewhere (e.g. exported from the JVM)."""
        cls.alpha = np.ascontiguousarray(alpha, np.float64).reshape(tables * divisions, m, dimension)
        cls.r = np.ascontiguousarray(r, np.float64).reshape(tables * divisions, m)
        cls.omega = np.ascontiguousarray(omega, np.float64).reshape(tables * divisions, m)
        if not np.all(cls.omega > 0):
            raise N.FspannArgumentError("omega_j <= 0")
        cls.DIM, cls.M, cls.LAMBDA, cls.BASE_SEED, cls.TABLES, cls.DIVISIONS = dimension, m, lambda_, baseSeed, tables, divisions
        cls._init = True

    @classmethod
    def isInitialized(cls): return cls._init

    @classmethod
    def reset(cls):
        cls._init = False
        cls.DIM = cls.M = cls.LAMBDA = cls.TABLES = cls.DIVISIONS = -1
        cls.BASE_SEED = -1
        cls.alpha = cls.r = cls.omega = None

    @classmethod
    def getStats(cls):
        return dict(initialized=cls._init, dimension=cls.DIM, m=cls.M, tables=cls.TABLES, divisions=cls.DIVISIONS,
                    **{"lambda": cls.LAMBDA},
                    omegaMin=float(cls.omega.min()) if cls._init else None,
                    omegaMax=float(cls.omega.max()) if cls._init else None,
                    omegaMean=float(cls.omega.mean()) if cls._init else None)

    @classmethod
    def get(cls, dimension, table, division):
        if not cls._init:
            raise N.FspannStateError("GFunctionRegistry not initialized")
        if dimension != cls.DIM:
            raise N.FspannArgumentError(f"Dimension mismatch: expected {cls.DIM}, got {dimension}")
        if not (0 <= table < cls.TABLES and 0 <= division < cls.DIVISIONS):
            raise N.FspannStateError(f"Missing GFunction for table={table}, division={division}")
        td = table * cls.DIVISIONS + division
        return cls.alpha[td], cls.r[td], cls.omega[td]


# --------------------------------------------------------------------------------------
# PartitionedIndexService — Setup + Route
# --------------------------------------------------------------------------------------
class PartitionedIndexService:
    DEFAULT_MAX_PROBES = 5

    def __init__(self, metadata, cfg: SystemConfig, keyService, cryptoService, device: int = 0):
        for name, v in (("metadata", metadata), ("cfg", cfg), ("keyService", keyService), ("cryptoService", cryptoService)):
            if v is None:
                raise N.FspannNullError(name)
        self.metadata, self.cfg, self.keyService, self.cryptoService = metadata, cfg, keyService, cryptoService
        self.device = device
        self.ctx: Optional[FspannContext] = None
        self.dim: Optional[int] = None
        self._frozen = False
        self._sample: List[np.ndarray] = []
        self._pending: List[tuple] = []      # (id, vector) parked before registry init (PIS:292-298)
        self._staged_ids: List[str] = []     # ids in the order they reach `staged`
        self._staged_vecs: List[np.ndarray] = []
        self._handle: Dict[str, int] = {}
        self._ids: List[str] = []
        self._probeOverride = -1
        self._lastRaw = 0
        self._lastTouched: List[str] = []

    # ---- Setup ----------------------------------------------------------------------
    def _initializeRegistry(self):
        pc = self.cfg
        if len(self._sample) < MIN_SAMPLE_SIZE:
            raise N.FspannStateError(f"Refusing to initialize GFunctionRegistry with sampleSize={len(self._sample)}")
        dim = len(self._sample[0])
        self._ensure_ctx(dim)
        GFunctionRegistry.initialize(np.stack(self._sample), dim, pc.m, pc.lambda_, pc.seed, pc.tables, pc.divisions, ctx=self.ctx)
        self._sample = []

    def _ensure_ctx(self, dim):
        if self.ctx is None:
            self.dim = dim
            self.ctx = FspannContext(self.cfg.native(dim), self.device)

    def insert(self, id: str, vector):  # PIS:265-312
        if id is None:
            raise N.FspannNullError("id cannot be null")
        if vector is None:
            raise N.FspannNullError("vector cannot be null")
        vector = np.asarray(vector, dtype=np.float64)
        if GFunctionRegistry.isInitialized():
            if len(vector) != GFunctionRegistry.DIM:
                raise N.FspannArgumentError(
                    f"Mixed dimensions not supported in single index: got {len(vector)}, expected {GFunctionRegistry.DIM}")
        else:
            if len(self._sample) < MAX_SAMPLE_SIZE:
                self._sample.append(vector.copy())
            if len(self._sample) >= MIN_SAMPLE_SIZE:
                self._initializeRegistry()
        if not GFunctionRegistry.isInitialized():
            self._pending.append((id, vector.copy()))
            return
        ep = self.cryptoService.encrypt(id, vector, self.keyService.getCurrentVersion())
        self._stage(ep, vector)

    def _stage(self, ep, vec):  # PIS:314-347 / directInsert :759-787 (codes are computed in bulk at finalize)
        self.metadata.saveEncryptedPoint(ep)
        if ep.id in self._handle:       # HashMap.put of an existing key: position kept, code replaced
            self._staged_vecs[self._staged_ids.index(ep.id)] = vec
            return
        self._handle[ep.id] = len(self._ids)
        self._ids.append(ep.id)
        self._staged_ids.append(ep.id)
        self._staged_vecs.append(vec)

    def finalizeForSearch(self):  # PIS:789-845
        if self._frozen:
            return
        if not GFunctionRegistry.isInitialized():
            if len(self._sample) >= MIN_SAMPLE_SIZE:
                self._initializeRegistry()
            else:
                raise N.FspannStateError(f"Cannot finalize index: only {len(self._sample)} samples collected (< MIN_SAMPLE_SIZE)")
        st, pc = GFunctionRegistry.getStats(), self.cfg
        if st["m"] != pc.m or st["lambda"] != pc.lambda_ or st["tables"] != pc.tables or st["divisions"] != pc.divisions:
            raise N.FspannStateError(f"GFunctionRegistry mismatch at finalize: {st}")
        for pid, vec in self._pending:
            self._stage(self.cryptoService.encrypt(pid, vec), vec)
        self._pending = []
        if self._staged_ids:
            dim = len(self._staged_vecs[0])
            self._ensure_ctx(dim)
            self.ctx.set_gfunctions(GFunctionRegistry.alpha, GFunctionRegistry.r, GFunctionRegistry.omega)
            n = len(self._ids)
            jh = np.array([_java_string_hash(s) for s in self._ids], dtype=np.int32)
            dl = np.array([1 if self.metadata.isDeleted(s) else 0 for s in self._ids], dtype=np.uint8)
            self.ctx.set_id_meta(n, jh, dl if dl.any() else None)
            # handles were assigned in staged order, so order == identity over the staged list
            self.ctx.build_index(np.stack(self._staged_vecs), order=np.arange(n, dtype=np.int32))
            self._staged_vecs = []
        self._frozen = True

    def refreshDeleted(self):
        """Re-read metadata.isDeleted for every id (the reference asks RocksDB per id per query, PIS:739)."""
        if self.ctx is not None and self._ids:
            dl = np.array([1 if self.metadata.isDeleted(s) else 0 for s in self._ids], dtype=np.uint8)
            jh = np.array([_java_string_hash(s) for s in self._ids], dtype=np.int32)
            self.ctx.set_id_meta(len(self._ids), jh, dl if dl.any() else None)

    # ---- Route ------------------------------------------------------------------------
    def _checkToken(self, token):
        if token is None:
            raise N.FspannNullError("token")
        if not self._frozen:
            raise N.FspannStateError("Index not finalized")
        if self.ctx is None or token.getDimension() != self.dim:
            return None  # dims.get(dim) == null -> List.of()  (PIS:598)
        q = token.getBitCodes()
        if q is None:
            raise N.FspannStateError("MSANNP violation: QueryToken missing BitSet codes")
        if q.shape[0] != self.cfg.tables:
            raise N.FspannStateError(f"Token tables mismatch: token={q.shape[0]} index={self.cfg.tables}")
        if q.shape[1] < self.cfg.divisions:
            raise N.FspannStateError(f"Token divisions mismatch at table=0 expectedDivisions>={q.shape[1] + 1}")
        return q[:, :self.cfg.divisions]

    def _route(self, codes, limit):
        res = self.ctx.route(codes[None], probe_override=self._probeOverride, limit=limit)
        n = int(res["count"][0])
        self._lastRaw = int(res["raw_seen"][0])
        ids = [self._ids[h] for h in res["ids"][0, :n]]
        return ids, res["score"][0, :n], int(res["kept"][0])

    def _route_batch(self, codes_list, limit, probe_override):
        """One fspann_route for a batch of tokens: per token (ids, scores, kept, rawSeen)."""
        res = self.ctx.route(np.stack(codes_list), probe_override=probe_override, limit=limit)
        out = []
        for i in range(len(codes_list)):
            n = int(res["count"][i])
            out.append(([self._ids[h] for h in res["ids"][i, :n]], res["score"][i, :n], int(res["kept"][i]), int(res["raw_seen"][i])))
        return out

    def lookupCandidatesWithScores(self, token) -> List[CandidateWithScore]:  # PIS:592-715
        codes = self._checkToken(token)
        if codes is None:
            return []
        ids, score, _ = self._route(codes, N.INT32_MAX)
        self._lastTouched = ids
        return [CandidateWithScore(i, int(s)) for i, s in zip(ids, score)]

    def lookupCandidateIds(self, token) -> List[str]:  # PIS:459-582 (truncated to HARD_CAP, :558-565)
        codes = self._checkToken(token)
        if codes is None:
            return []
        ids, _, _ = self._route(codes, max(self.cfg.maxGlobalCandidates, self.cfg.refinementLimit))
        self._lastTouched = ids
        return ids

    def loadPointIfActive(self, id):  # PIS:717-724
        if self.metadata.isDeleted(id):
            return None
        try:
            return self.metadata.loadEncryptedPoint(id)
        except Exception:
            return None

    def isFrozen(self): return self._frozen
    def numTables(self): return self.cfg.tables
    def getDefaultMaxProbes(self): return self.DEFAULT_MAX_PROBES
    def setProbeOverride(self, probes): self._probeOverride = int(probes)
    def clearProbeOverride(self): self._probeOverride = -1
    def getLastRawCandidateCount(self): return self._lastRaw
    def getLastTouchedIds(self): return frozenset(self._lastTouched)
    def getLastTouchedCount(self): return len(self._lastTouched)
    def handleOf(self, id): return self._handle[id]
    def idOf(self, handle): return self._ids[handle]


def _java_string_hash(s: str) -> int:
    """String.hashCode over UTF-16 code units."""
    b = s.encode("utf-16-be")
    h = 0
    for cu in struct.unpack(">%dH" % (len(b) // 2), b):
        h = (31 * h + cu) & 0xFFFFFFFF
    return h - (1 << 32) if h & 0x80000000 else h


# --------------------------------------------------------------------------------------
# QueryTokenFactory — TokenGen
# --------------------------------------------------------------------------------------
class QueryTokenFactory:
    def __init__(self, crypto, keyService, cfg: SystemConfig, ctx: FspannContext = None):
        for v in (crypto, keyService, cfg):
            if v is None:
                raise N.FspannNullError("QueryTokenFactory dependency")
        self.crypto, self.keyService, self.cfg = crypto, keyService, cfg
        self._ctx = ctx

    def _context(self, dim):
        if self._ctx is None or self._ctx.cfg.dim != dim:
            self._ctx = FspannContext(self.cfg.native(dim))
            self._ctx.set_gfunctions(GFunctionRegistry.alpha, GFunctionRegistry.r, GFunctionRegistry.omega)
            self._g_id = id(GFunctionRegistry.alpha)
        elif getattr(self, "_g_id", None) != id(GFunctionRegistry.alpha):
            self._ctx.set_gfunctions(GFunctionRegistry.alpha, GFunctionRegistry.r, GFunctionRegistry.omega)
            self._g_id = id(GFunctionRegistry.alpha)
        return self._ctx

    def create(self, vec, topK: int) -> QueryToken:  # QueryTokenFactory.java:63-167
        return self.createBatch([vec], topK)[0]

    def createBatch(self, vecs, topK: int) -> List[QueryToken]:
        if vecs is None or any(v is None for v in vecs):
            raise N.FspannNullError("query vector is null")
        if topK <= 0:
            raise N.FspannArgumentError("topK must be > 0")
        if not GFunctionRegistry.isInitialized():
            raise N.FspannStateError("GFunctionRegistry not initialized. Build index first.")
        pc = self.cfg
        vecs = [np.asarray(v, dtype=np.float64) for v in vecs]
        dim = len(vecs[0])
        st = GFunctionRegistry.getStats()
        if any(len(v) != dim for v in vecs) or st["dimension"] != dim or st["tables"] != pc.tables \
                or st["divisions"] != pc.divisions or st["m"] != pc.m or st["lambda"] != pc.lambda_:
            raise N.FspannStateError(f"GFunctionRegistry mismatch: {st}")
        ctx = self._context(dim)
        codes = ctx.encode(np.stack(vecs))  # NaN/Inf -> FspannArgumentError("Vector contains NaN/Inf")
        codes = codes.reshape(len(vecs), pc.tables, pc.divisions, -1)
        kv = self.keyService.getCurrentVersion()
        out = []
        for v, bc in zip(vecs, codes):
            iv = os.urandom(12)
            ct = self.crypto.encryptQuery(v, kv.key, iv)
            out.append(QueryToken(bc, iv, ct, topK, pc.tables, dim, kv.version, pc.lambda_, f"dim_{dim}_v{kv.version}"))
        return out

    def derive(self, tok: QueryToken, newTopK: int) -> QueryToken:  # :182-198
        if tok is None:
            raise N.FspannNullError("token is null")
        if newTopK <= 0:
            raise N.FspannArgumentError("newTopK must be > 0")
        return QueryToken(tok.getBitCodes(), tok.getIv(), tok.getEncryptedQuery(), newTopK, tok.getNumTables(),
                          tok.getDimension(), tok.getVersion(), tok.getLambda(), tok.getEncryptionContext())


# --------------------------------------------------------------------------------------
# QueryServiceImpl — Refine
# --------------------------------------------------------------------------------------
class QueryServiceImpl:
    def __init__(self, index: PartitionedIndexService, cryptoService, keyService, tf: Optional[QueryTokenFactory], cfg: SystemConfig):
        for name, v in (("index", index), ("cryptoService", cryptoService), ("keyService", keyService), ("cfg", cfg)):
            if v is None:
                raise N.FspannNullError(name)
        self.index, self.cryptoService, self.keyService, self.tokenFactory, self.cfg = index, cryptoService, keyService, tf, cfg
        self._refineOverride = None
        self.reencTracker = None
        self._clear()

    def _clear(self):
        self.lastCandTotal = self.lastCandKept = self.lastCandDecrypted = self.lastReturned = 0
        self.lastCandIds: List[str] = []
        self.lastUniqueCandidates = 0
        self.touchedThisSession = set()

    def search(self, token: Optional[QueryToken]) -> List[QueryResult]:
        return self.searchBatch([token])[0]

    def searchBatch(self, tokens: Sequence[Optional[QueryToken]]) -> List[List[QueryResult]]:
        """QSI.search (QSI:101-352) for a batch; metrics (`getLast*`) describe the LAST token."""
        return [self._search_one(t) for t in tokens] if len(tokens) <= 1 else self._search_many(tokens)

    # single-token path = literal control flow of the reference
    def _search_one(self, token):
        if token is None:
            return []
        self._clear()
        try:
            qkv = self.keyService.getVersion(token.getVersion())
        except Exception:
            qkv = self.keyService.getCurrentVersion()
        qVec = np.asarray(self.cryptoService.decryptQuery(token.getEncryptedQuery(), token.getIv(), qkv.key), np.float64)
        if not np.all(np.isfinite(qVec)):
            return []
        idx, K = self.index, token.getTopK()
        retried = False
        try:
            while True:
                codes = idx._checkToken(token)
                if codes is None:
                    return []
                runtimeLimit = self.getEffectiveRefinementLimit(self.cfg.refinementLimit)
                ids, _score, kept = idx._route(codes, runtimeLimit)           # stage A + A.5 on the GPU
                idx._lastTouched = ids
                self.lastCandTotal, self.lastCandKept = idx.getLastRawCandidateCount(), kept
                if kept == 0:
                    return []
                self.lastUniqueCandidates = len(ids)
                rows, rids = [], []
                for cid in ids:                                               # stage B host part (QSI:238-271)
                    try:
                        ep = idx.loadPointIfActive(cid)
                        if ep is None:
                            continue
                        v = np.asarray(self.cryptoService.decryptFromPoint(ep, self.keyService.getVersion(ep.version).key), np.float64)
                        if v.shape != qVec.shape or not np.all(np.isfinite(v)):
                            continue
                        rows.append(v)
                        rids.append(cid)
                        self.touchedThisSession.add(cid)
                    except Exception:
                        continue
                self.lastCandDecrypted = len(rows)
                if not rows:
                    return []
                B = len(rows)
                packed = idx.ctx.host_buffer((1, B, len(qVec)), np.float64)     # the context's pinned block (fspann_host_buffer): one DMA
                packed[0] = np.stack(rows)
                res = idx.ctx.refine(qVec[None], packed, np.arange(B, dtype=np.int32)[None],
                                     np.array([B], np.int32), K)          # stage B distances + C on the GPU
                eff = int(res["count"][0])
                out = [QueryResult(rids[j], float(dd)) for j, dd in zip(res["ids"][0, :eff], res["dist"][0, :eff])]
                self.lastReturned, self.lastCandIds = eff, [r.id for r in out]
                if not retried and (self.lastReturned < K or self.lastCandDecrypted < 10 * K):  # QSI:327-337,444-447
                    retried = True
                    idx.setProbeOverride(10)
                    continue
                return out
        finally:
            idx.clearProbeOverride()
            if self.reencTracker is not None and self.touchedThisSession:
                self.reencTracker.record(set(self.touchedThisSession))

    def _search_many(self, tokens):
        """The batched mirror (GpuQueryServiceImpl.searchBatch): the reference calls search(token) once per query from one serial
        loop (ForwardSecureANNSystem.java:636-748); here the tokens of a batch share ONE fspann_route, the host decrypt loop
        (QSI:238-271, unchanged) runs over every F_q, and ONE fspann_refine scores all of them — then the adaptive retry
        (QSI:327-337) reruns, again as one batch, exactly the queries that came back short.  Results per token are those of
        search(token); the metric getters describe the LAST token, `touchedThisSession` is per token as in the reference."""
        idx = self.index
        n = len(tokens)
        results: List[List[QueryResult]] = [[] for _ in range(n)]
        Ks = {t.getTopK() for t in tokens if t is not None}
        if len(Ks) != 1:                          # mixed topK: fspann_refine takes one k per call
            return [self._search_one(t) for t in tokens]
        K = Ks.pop()
        qvecs, codes = {}, {}
        for i, t in enumerate(tokens):            # QSI:102-140 per token
            if t is None:
                continue
            try:
                qkv = self.keyService.getVersion(t.getVersion())
            except Exception:
                qkv = self.keyService.getCurrentVersion()
            qv = np.asarray(self.cryptoService.decryptQuery(t.getEncryptedQuery(), t.getIv(), qkv.key), np.float64)
            if not np.all(np.isfinite(qv)):
                continue
            c = idx._checkToken(t)
            if c is None:
                continue
            qvecs[i], codes[i] = qv, c
        metrics = {i: dict(total=0, kept=0, decrypted=0, returned=0, unique=0, ids=[], touched=set()) for i in range(n)}
        active = sorted(qvecs)
        probe = idx._probeOverride
        limit = self.getEffectiveRefinementLimit(self.cfg.refinementLimit)
        try:
            for attempt in range(2):
                if not active:
                    break
                routed = idx._route_batch([codes[i] for i in active], limit, probe)            # stage A + A.5: one call
                rows_all, rids_all = {}, {}
                for i, (ids, _score, kept, raw) in zip(active, routed):
                    m = metrics[i]
                    m["total"], m["kept"], m["unique"] = raw, kept, len(ids)
                    rows, rids = [], []
                    for cid in ids:                                                           # stage B host part (QSI:238-271)
                        try:
                            ep = idx.loadPointIfActive(cid)
                            if ep is None:
                                continue
                            v = np.asarray(self.cryptoService.decryptFromPoint(ep, self.keyService.getVersion(ep.version).key), np.float64)
                            if v.shape != qvecs[i].shape or not np.all(np.isfinite(v)):
                                continue
                            rows.append(v)
                            rids.append(cid)
                            m["touched"].add(cid)
                        except Exception:
                            continue
                    m["decrypted"] = len(rows)
                    rows_all[i], rids_all[i] = rows, rids
                    if not rows:
                        results[i], m["returned"], m["ids"] = [], 0, []
                scored = [i for i in active if rows_all[i]]
                if scored:
                    Bm = max(len(rows_all[i]) for i in scored)
                    dim = len(qvecs[scored[0]])
                    cand = idx.ctx.host_buffer((len(scored), Bm, dim), np.float64)   # pinned; rows beyond a query's count are never read as valid
                    cnt = np.zeros(len(scored), np.int32)
                    for j, i in enumerate(scored):
                        cand[j, :len(rows_all[i])] = np.stack(rows_all[i])
                        cnt[j] = len(rows_all[i])
                    res = idx.ctx.refine(np.stack([qvecs[i] for i in scored]), cand, np.tile(np.arange(Bm, dtype=np.int32), (len(scored), 1)),
                                         cnt, K)                                              # stage B distances + C: one call
                    for j, i in enumerate(scored):
                        eff = int(res["count"][j])
                        out = [QueryResult(rids_all[i][r], float(dd)) for r, dd in zip(res["ids"][j, :eff], res["dist"][j, :eff])]
                        results[i] = out
                        metrics[i]["returned"], metrics[i]["ids"] = eff, [r.id for r in out]
                if attempt == 0:                   # QSI:327-337,444-447: one more pass with 10 probes for the short ones
                    active = [i for i in active if rows_all[i] and (metrics[i]["returned"] < K or metrics[i]["decrypted"] < 10 * K)]
                    probe = 10
        finally:
            idx.clearProbeOverride()
        self._clear()
        last = max(metrics) if metrics else None
        for i in range(n):
            if self.reencTracker is not None and metrics[i]["touched"]:
                self.reencTracker.record(set(metrics[i]["touched"]))
        if last is not None:
            m = metrics[last]
            self.lastCandTotal, self.lastCandKept, self.lastCandDecrypted = m["total"], m["kept"], m["decrypted"]
            self.lastReturned, self.lastCandIds, self.lastUniqueCandidates = m["returned"], m["ids"], m["unique"]
            self.touchedThisSession = set(m["touched"])
        return results

    # metrics / overrides (QSI:417-474)
    def getLastCandTotal(self): return self.lastCandTotal
    def getLastCandKept(self): return self.lastCandKept
    def getLastCandDecrypted(self): return self.lastCandDecrypted
    def getLastReturned(self): return self.lastReturned
    def getLastFinalResultIds(self): return list(self.lastCandIds)
    def getLastUniqueCandidates(self): return self.lastUniqueCandidates
    def setRefinementLimit(self, limit): self._refineOverride = int(limit)
    def clearRefinementLimit(self): self._refineOverride = None
    def getEffectiveRefinementLimit(self, defaultLimit):
        return self._refineOverride if (self._refineOverride is not None and self._refineOverride > 0) else defaultLimit
    def setReencryptionTracker(self, tr): self.reencTracker = tr
    def deriveToken(self, base, k):
        if self.tokenFactory is None:
            raise N.FspannStateError("QueryTokenFactory not available")
        return self.tokenFactory.derive(base, k)
