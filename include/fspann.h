/* =============================================================================
 * fspann.h — C ABI of libfspann_hip.so: MI355X-native (gfx950) TokenGen -> Route
 * -> Refine for FSPANN.
 *
 * This is the drop-in boundary for the ONE hot path of
 * Mehran-Memon/fspann-query-system.  The reference is pure Java with no FFI seam
 * (constructor wiring, ForwardSecureANNSystem.java:316-322,362-398), so each entry
 * point below names the reference method whose arithmetic it replaces; the JNI stub
 * a maintainer adds is shown in INTEGRATION.md and jni/fspann_jni.cpp.
 *
 * Paths (under /root/reference/fsp-anns-parent/):
 *   idx = index/src/main/java/com/fspann/index/paper
 *   qry = query/src/main/java/com/fspann/query
 *   PIS = idx/PartitionedIndexService.java      QSI = qry/service/QueryServiceImpl.java
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types cross this boundary.
 *   - every call returns 0 (FSPANN_OK) or a negative error mapped 1:1 onto the Java
 *     exception class the reference would throw; fspann_last_error() returns a
 *     thread-local message.  No C++ exception crosses the ABI.
 *   - buffers are caller-owned.  Functions WITHOUT the _dev suffix take host
 *     pointers and copy in/out; functions WITH _dev take device (HBM) pointers and
 *     only enqueue work on the context's HIP stream (fspann_ctx_stream); the caller
 *     synchronises with fspann_sync() or its own stream/event calls.
 *   - ids are int32 handles.  The Java adapter keeps String id <-> handle and
 *     hands String.hashCode() per handle to fspann_set_id_meta (it decides the
 *     reference's HashMap iteration order, see DESIGN.md "Java order key").
 *   - one context = one GPU = one stream; calls on a context are serialised INSIDE
 *     the library (a per-context lock held for the duration of every entry point), so
 *     a context shared between threads is safe, just not parallel — use
 *     fspann_ctx_clone for that (QueryServiceImpl is not re-entrant either, QSI:45-64).
 * ========================================================================== */
#ifndef FSPANN_H
#define FSPANN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSPANN_OK 0
#define FSPANN_E_STATE (-1)  /* java.lang.IllegalStateException   (PIS:594,602,605,630; QueryTokenFactory.java:67-88) */
#define FSPANN_E_ARG (-2)    /* java.lang.IllegalArgumentException (Coding.java:355-361; QueryTokenFactory.java:65)  */
#define FSPANN_E_NULL (-3)   /* java.lang.NullPointerException     (Objects.requireNonNull)                          */
#define FSPANN_E_DEVICE (-4) /* HIP runtime failure                                                                   */
#define FSPANN_E_NOMEM (-5)  /* allocation failure                                                                    */
#define FSPANN_E_RANGE (-6)  /* a caller-provided capacity is too small                                               */

#define FSPANN_F32 0
#define FSPANN_F64 1

typedef struct fspann_ctx fspann_ctx;

/* paper.* and runtime.* knobs of config/SystemConfig.java:237-338 that the path reads. */
typedef struct fspann_cfg {
    int32_t tables;                 /* paper.tables   (T)                                   */
    int32_t divisions;              /* paper.divisions (D)                                  */
    int32_t m;                      /* paper.m        projections per GFunction             */
    int32_t lambda;                 /* paper.lambda   bits per projection                   */
    int32_t dim;                    /* vector dimension d                                   */
    int32_t block_size;             /* PIS:92  DEFAULT_GREEDY_BLOCK_SIZE = 64 (0 => 64)     */
    int32_t default_probes;         /* PIS:93  DEFAULT_MAX_PROBES = 5        (0 => 5)       */
    int32_t probe_override;         /* runtime.probeOverride (<= 0: none), PIS:884-885      */
    int32_t max_global_candidates;  /* runtime.maxGlobalCandidates (0 => 20000)             */
    int32_t refinement_limit;       /* runtime.refinementLimit    (0 => 20000)              */
    int32_t hamming_prefilter_threshold; /* runtime.hammingPrefilterThreshold (QSI:169)     */
    int32_t reserved;
} fspann_cfg;

/* ---- lifecycle ----------------------------------------------------------------- */
int fspann_ctx_create(int device, const fspann_cfg* cfg, fspann_ctx** out);
void fspann_ctx_destroy(fspann_ctx* ctx);
/* A second context on the same device that READS src's GFunctions, frozen index, id metadata and store in place — no
 * second copy in HBM, one working set in the caches — and owns its HIP stream and work areas: the way to serve one index from
 * several threads / streams (a context is not re-entrant; the reference shares one PartitionedIndexService between its
 * query threads the same way, PIS:68-73).  src must be finalized.  While clones are alive the shared state is read-only
 * (the mutating entry points return FSPANN_E_STATE on the owner and on the clones); destroying the owner first is allowed,
 * its arrays are released with the last clone.                                                                            */
int fspann_ctx_clone(fspann_ctx* src, fspann_ctx** out);
const char* fspann_last_error(void);
const char* fspann_version(void);
/* hipStream_t of the context, as void* (for hipEvent timing / torch.cuda.ExternalStream). */
void* fspann_ctx_stream(fspann_ctx* ctx);
int fspann_sync(fspann_ctx* ctx);

/* ---- Setup: import the frozen routing state ----------------------------------------
 * Replaces the JVM-heap state of GFunctionRegistry (idx/GFunctionRegistry.java:63-147) and
 * PIS.dims (PIS:60,96-113).  alpha[T*D][m][dim], r[T*D][m], omega[T*D][m], all fp64, exactly
 * the Coding.GFunction fields (idx/Coding.java:52-97).  (t,d) is flattened td = t*D + d.   */
int fspann_set_gfunctions(fspann_ctx* ctx, const double* alpha, const double* r, const double* omega);

/* Native GFunctionRegistry.initialize (idx/GFunctionRegistry.java:63-147 -> Coding.buildFromSample,
 * idx/Coding.java:184-241): sample = [ns][dim] fp64 (the reference passes the first 1000 inserted
 * vectors, PIS:280-289); seed(t,d) = base_seed + t*1000003 + d.  alpha comes from Box-Muller on
 * SplittableRandom with the platform libm, so it is self-consistent but NOT claimed equal to a
 * JVM's alpha for the same seed — a Java caller imports its own with fspann_set_gfunctions.   */
int fspann_registry_initialize(fspann_ctx* ctx, const double* sample, int64_t ns, int64_t base_seed);
int fspann_get_gfunctions(fspann_ctx* ctx, double* alpha, double* r, double* omega);

/* One (t,d) table of GreedyPartitioner.Partition (idx/GreedyPartitioner.java:13-32) as SoA:
 * min_key/max_key[n_parts], rep[n_parts][W] (W = ceil(m*lambda/64) BitSet words, bit i ->
 * word i/64 bit i%64), id_off[n_parts+1], ids[id_off[n_parts]] in partition order.          */
int fspann_set_index(fspann_ctx* ctx, int td, int64_t n_parts, const int64_t* min_key, const int64_t* max_key,
                     const uint64_t* rep, const int64_t* id_off, const int32_t* ids);

/* java_hash[h] = String.hashCode() of the id behind handle h (NULL: ids are the decimal
 * ordinals Long.toString(h), ForwardSecureANNSystem.java:515); deleted[h] != 0 mirrors
 * metadata.isDeleted(id) (PIS:739; common/RocksDBMetadataManager.java:203-224); NULL: none. */
int fspann_set_id_meta(fspann_ctx* ctx, int64_t n_ids, const int32_t* java_hash, const uint8_t* deleted);

/* PIS.finalizeForSearch (PIS:789-845): freezes the context; Route calls before it fail
 * with FSPANN_E_STATE ("Index not finalized", PIS:594).                                     */
int fspann_finalize(fspann_ctx* ctx);

/* Native Setup (replaces PIS.insert's coding loop PIS:331-346 + PIS.build
 * PIS:372-434 + GreedyPartitioner.build): codes all n vectors on the GPU (MFMA fp32 pre-filter + exact fp64
 * re-check for n >= 4096, the exact fp64 kernel otherwise: bit-identical codes either way) and cuts the partitions on
 * the GPU too (stable radix sort by key of the HashMap iteration order, csrc/build.hip.h).  `order[n]` = handles in the order they reach `staged`
 * (NULL: the stock pipeline's 999,1000,..,n-1,0,..,998).  vectors = [n][dim]
 * host, row h = handle h.  Requires set_gfunctions + set_id_meta first.  Freezes ctx.     */
int fspann_build_index(fspann_ctx* ctx, int64_t n, const void* vectors, int dtype, const int32_t* order);

/* The same Setup with the rows handed over in PIECES: IndexService.insert(String, double[]) is one vector at a time
 * (common/.../IndexService.java:19, PIS:265-347) and a JVM's direct buffers hold at most 2 GB, while config #3 / #4 hand over
 * 7.7 / 30.7 GB of rows.  begin(n_hint) -> append(rows of the next n_rows handles, fp32 or fp64; any chunk size) ... ->
 * finish(order): every chunk is coded on arrival and only its codes (T*D*W*8 bytes per row) stay in HBM; finish cuts the
 * partitions over the rows appended and freezes the context exactly as fspann_build_index does (which is begin + one append +
 * finish).  n_hint sizes the code buffer (the exact total avoids a regrow; more rows than hinted are accepted).  Needs
 * fspann_set_gfunctions before begin and fspann_set_id_meta (for >= the rows appended) before finish.               */
int fspann_build_begin(fspann_ctx* ctx, int64_t n_hint);
int fspann_build_append(fspann_ctx* ctx, int64_t n_rows, const void* rows, int dtype);
int fspann_build_finish(fspann_ctx* ctx, const int32_t* order);

/* Live mirror of metadata.isDeleted(id) (PIS:739: asked for every id of every probed partition at query time, so a delete shows in
 * the next query).  Sets (flag != 0) or clears the deleted bit of `handles[0..n)` WITHOUT un-freezing the context; may be called
 * on the index owner or on any of its clones while all of them serve queries: Route calls enqueued after it returns see the
 * change, calls in flight see the old or the new flag (as a JVM query racing a delete does).                      */
int fspann_set_deleted(fspann_ctx* ctx, const int32_t* handles, int64_t n, int flag);

/* Frozen-index file: GFunctions + id metadata + every table as flat little-endian SoA.  The reference
 * never persists routing state and rebuilds it by decrypting every point (ForwardSecureANNSystem.java:926-948).
 * load() requires a context created with the same tables/divisions/m/lambda/dim and freezes it.              */
int fspann_index_save(fspann_ctx* ctx, const char* path);
int fspann_index_load(fspann_ctx* ctx, const char* path);

/* Export a built/imported table (sizes via fspann_index_dims). */
int fspann_index_dims(fspann_ctx* ctx, int td, int64_t* n_parts, int64_t* n_ids);
int fspann_get_index(fspann_ctx* ctx, int td, int64_t* min_key, int64_t* max_key, uint64_t* rep,
                     int64_t* id_off, int32_t* ids);

/* ---- TokenGen math: Coding.H + Coding.C for all T*D GFunctions ------------------------
 * Replaces the loop QueryTokenFactory.create :98-131 (and PIS.insert :331-346).
 * codes = [nq][T*D][W].  Exact: sequential fp64, no FMA == Java (idx/Coding.java:250-258,
 * 285-301, 349-353).  NaN/Inf in a vector => FSPANN_E_ARG ("Vector contains NaN/Inf").
 * hashes (optional, may be NULL) = [nq][T*D][m] int32 Coding.H values.                    */
int fspann_encode(fspann_ctx* ctx, int64_t nq, const void* q, int dtype, uint64_t* codes, int32_t* hashes);
int fspann_encode_dev(fspann_ctx* ctx, int64_t nq, const void* q_dev, int dtype, uint64_t* codes_dev,
                      int32_t* hashes_dev, int32_t* bad_dev /* [nq], 1 where NaN/Inf */);

/* Encode path: 0 = auto (MFMA fp32 GEMM pre-filter + exact fp64 re-check for nq >= 4096, exact fp64 VALU kernel
 * otherwise), 1 = exact only, 2 = always MFMA + re-check.  Every mode returns bit-identical hashes/codes: an
 * MFMA result is kept only when its error interval lies strictly inside one bucket of floor((y+r)/omega).    */
int fspann_set_encode_mode(fspann_ctx* ctx, int mode);
int64_t fspann_last_encode_rechecked(fspann_ctx* ctx);

/* ---- Route: PIS.lookupCandidatesWithScores (PIS:592-715) ---------------------------
 * codes = [nq][T*D][W].  probe_override > 0 overrides like PIS.setProbeOverride (PIS:868,
 * 880-888).  Writes, per query, the first min(limit, kept) entries of the reference's result
 * list (HashMap iteration order stable-sorted by score) into ids/score[nq][cap];
 * count[q] = entries written, kept[q] = size of the full list (QSI.lastCandKept),
 * raw_seen[q] = PIS.getLastRawCandidateCount().
 *   limit = INT32_MAX       -> lookupCandidatesWithScores (full list)
 *   limit = HARD_CAP        -> lookupCandidateIds (PIS:459-582, truncation :558-565)
 *   limit = refinementLimit -> QSI stage A.5 (QSI:169-214), F_q
 * cap < min(limit, worst case) => FSPANN_E_RANGE.                                          */
int fspann_route(fspann_ctx* ctx, int64_t nq, const uint64_t* codes, int probe_override, int32_t limit,
                 int64_t cap, int32_t* ids, int32_t* score, int32_t* count, int32_t* kept, int32_t* raw_seen);
int fspann_route_dev(fspann_ctx* ctx, int64_t nq, const uint64_t* codes_dev, int probe_override, int32_t limit,
                     int64_t cap, int32_t* ids_dev, int32_t* score_dev, int32_t* count_dev, int32_t* kept_dev,
                     int32_t* raw_seen_dev);
/* JDK HashMap order and its limit.  The list order above is the iteration order of HashMap<String,Long> bestScore
 * (PIS:619,690-693) stable-sorted by score; the library derives it in closed form (bin index at the map's final table
 * length, then first insertion) — exact while every bin is a plain chain.  A put that finds 8 nodes in its bin makes the
 * JVM treeify that bin, after which the closed form no longer holds.  That case is detected exactly on the GPU (per capacity
 * stage of the map): the query's count is set to -1 (Refine scores nothing for it) and it is handed to the host model below —
 * fspann_route finishes it before it returns; after the asynchronous _dev entry points fspann_unmodelled_queries reports how
 * many queries are flagged (synchronises) and fspann_route_resolve_dev finishes them.
 * fspann_build_index applies the same rule to HashMap<String,BitSet>(staged.size()) (PIS:413): when a bin of the staging map
 * treeifies, the map's iteration order comes from the host model instead of the GPU's (bin, position) sort.
 * The bounded select hands a query to the full select when >= 9 of the entries it holds share (score, bin); a bin that
 * reaches 9 ids only through candidates the bounded select never loads is not seen by it ("treeified bins"). */
int fspann_unmodelled_queries(fspann_ctx* ctx, int64_t* total, int reset);
/* Answer the flagged queries instead of refusing them.  A treeified bin's iteration order is java.util.HashMap.TreeNode's `next`
 * list (treeify keeps the chain order and moves the root to the front, putTreeVal links a new node behind its tree parent, resize
 * splits in order and untreeifies at <= 6): the library carries a literal host model of that (host/java_hashmap.hpp) and replays a
 * flagged query's lookupCandidatesWithScores put by put (host/route_replay.hpp) — the rare path, ~0.3 % of the queries at the
 * reference's shipped profiles.  fspann_route does this by itself.  After an asynchronous fspann_route_dev / fspann_tick_dev call,
 * fspann_route_resolve_dev (same arguments as that call) synchronises the stream, finishes every query whose count is -1 and
 * rewrites its ids / score / count / kept / raw_seen in place; *resolved = how many.  What stays -1 afterwards: a tree bin that
 * must order different ids with EQUAL String.hashCode (String.compareTo) when the ids are not decimal ordinals.              */
int fspann_route_resolve_dev(fspann_ctx* ctx, int64_t nq, const uint64_t* codes_dev, int probe_override, int32_t limit, int64_t cap,
                             int32_t* ids_dev, int32_t* score_dev, int32_t* count_dev, int32_t* kept_dev, int32_t* raw_seen_dev,
                             int64_t* resolved);
/* Worst-case entries per query for (probes): min(T*D*probes*block_size, HARD_CAP + block_size - 1). */
int64_t fspann_route_max_candidates(fspann_ctx* ctx, int probe_override);
int fspann_effective_probes(fspann_ctx* ctx, int probe_override); /* PIS:880-888 */
/* Select path of fspann_route: 0 = auto, 1 = always the full select, 2 = the bounded select whenever it is legal
 * (limit <= 1024, kept/raw_seen not requested, HARD_CAP and HashMap resize out of reach).  The bounded select reads
 * only the probed partitions with the smallest Hamming distances; every mode returns the identical list.        */
int fspann_set_route_mode(fspann_ctx* ctx, int mode);
/* Diagnostics: which select the last route call ran and how many queries the bounded select handed back. */
int fspann_last_route_info(fspann_ctx* ctx, int* lazy, int* overflowed);

/* ---- Refine: QSI stage B (distance part) + stage C ------------------------------------
 * Replaces QSI.l2 (QSI:364-372), isValid (:407-413), the stable sort + top-K (:298-316).
 * q = [nq][dim]; cand = [nq][B][dim] packed decrypted candidates (row j < cand_count[q] valid,
 * in F_q order; rows the host could not load/decrypt are simply absent); cand_ids[nq][B].
 * Distances are sequential fp64 like Java (bit-exact when inputs are exactly representable,
 * always <= 1e-5 relative).  Non-finite candidate rows are skipped; a non-finite query gives
 * out_count = 0.  out_ids/out_dist = [nq][k], padded with -1 / +inf; out_count[q] =
 * min(k, scored); scored[q] (optional) = QSI.lastCandDecrypted.                            */
int fspann_refine(fspann_ctx* ctx, int64_t nq, const void* q, const void* cand, int dtype, int64_t B,
                  const int32_t* cand_ids, const int32_t* cand_count, int k, int32_t* out_ids, double* out_dist,
                  int32_t* out_count, int32_t* scored);
/* Pinned host memory owned by the context, at least `bytes` long (grown on demand — an earlier pointer dies with a growth —, freed
 * with the context; NULL + fspann_last_error on failure): what the adapter packs the decrypted candidate rows into in place of
 * QSI's ArrayList<double[]> (QSI:238-271; GpuQueryServiceImpl wraps it in a direct ByteBuffer and hands that to fspann_refine).
 * Rows that start in pinned memory reach the GPU by plain DMA; rows in pageable memory go through the runtime's staging copies. */
void* fspann_host_buffer(fspann_ctx* ctx, size_t bytes);
int fspann_refine_dev(fspann_ctx* ctx, int64_t nq, const void* q_dev, int q_dtype, const void* cand_dev,
                      int cand_dtype, int64_t B, const int32_t* cand_ids_dev, const int32_t* cand_count_dev, int k,
                      int32_t* out_ids_dev, double* out_dist_dev, int32_t* out_count_dev, int32_t* scored_dev);

/* Measurement aid: between _begin and _end every refinement-scan dispatch of this context carries its own
 * start/stop HIP events (kernel-attached, on the context's stream); _end synchronises and returns the number of
 * timed dispatches and the sum of their durations.  Only every `every`-th dispatch is timed (an attached pair
 * costs a few microseconds of stream time); max_launches bounds the events kept.                          */
int fspann_refine_timing_begin(fspann_ctx* ctx, int max_launches, int every);
int fspann_refine_timing_end(fspann_ctx* ctx, int* launches, double* total_ms);

/* ---- plaintext store (TEST / BENCH harness only) ----------------------------------------
 * Stand-in for the host's loadPointIfActive + decryptFromPoint (PIS:717-724;
 * crypto/AesGcmCryptoService.java:126-166), which stay on the host in production: keeps
 * plaintext rows in HBM and packs F_q rows into the [nq][B][dim] buffer Refine consumes.   */
int fspann_store_set(fspann_ctx* ctx, int64_t n, const void* vectors, int dtype /* stored as given */);
/* The same over rows that already live in HBM (caller-owned, 16-byte aligned, [n][dim]): no copy; the caller keeps
 * them alive and unchanged until the next store_set / store_attach_dev / ctx_destroy.                        */
int fspann_store_attach_dev(fspann_ctx* ctx, int64_t n, const void* vectors_dev, int dtype);
int fspann_store_gather_dev(fspann_ctx* ctx, int64_t nq, const int32_t* sel_ids_dev, const int32_t* sel_count_dev,
                            int64_t B, void* cand_dev /* [nq][B][dim], store dtype */);
/* Refine with the candidate rows read from the resident store by id (row j of query q =
 * store[cand_ids[q][j]], j < cand_count[q]): the same scan + top-K as fspann_refine_dev without the
 * [nq][B][dim] staging copy.  An id outside [0, n) is a point that failed to load (QSI:252-256):
 * skipped, not scored.  FSPANN_E_STATE when no store has been set.                            */
int fspann_refine_store(fspann_ctx* ctx, int64_t nq, const void* q, int q_dtype, int64_t B, const int32_t* cand_ids,
                        const int32_t* cand_count, int k, int32_t* out_ids, double* out_dist, int32_t* out_count,
                        int32_t* scored);
int fspann_refine_store_dev(fspann_ctx* ctx, int64_t nq, const void* q_dev, int q_dtype, int64_t B,
                            const int32_t* cand_ids_dev, const int32_t* cand_count_dev, int k, int32_t* out_ids_dev,
                            double* out_dist_dev, int32_t* out_count_dev, int32_t* scored_dev);
/* QueryServiceImpl.search (QSI:101-352) for a batch whose candidate rows are resident in the store: encode ->
 * route(limit = B, counters not produced) -> refine_store, enqueued in stream order by one call.  The adaptive
 * retry (QSI:327-337) stays with the caller (out_count / scored say when).  Optional outputs may be NULL:
 * scored, sel_ids [nq][B] + sel_count [nq] (= F_q), bad [nq] (1 = query held NaN/Inf, QueryTokenFactory rejects). */
int fspann_search_store_dev(fspann_ctx* ctx, int64_t nq, const void* q_dev, int q_dtype, int probe_override, int64_t B,
                            int k, int32_t* out_ids_dev, double* out_dist_dev, int32_t* out_count_dev, int32_t* scored_dev,
                            int32_t* sel_ids_dev, int32_t* sel_count_dev, int32_t* bad_dev);
/* Completes the fspann_search_store_dev call that precedes it on this context (same arguments): synchronises, finishes the
 * queries Route flagged (count -1, see fspann_route_resolve_dev) and, when there were any, scores the batch again.            */
int fspann_search_store_finish_dev(fspann_ctx* ctx, int64_t nq, const void* q_dev, int q_dtype, int probe_override, int64_t B, int k,
                                   int32_t* out_ids_dev, double* out_dist_dev, int32_t* out_count_dev, int32_t* scored_dev,
                                   int32_t* sel_ids_dev, int32_t* sel_count_dev, int64_t* resolved);
const void* fspann_store_dev_ptr(fspann_ctx* ctx, int* dtype);

/* ---- one launch for three stages of three batches in flight --------------------------------------------------------------
 * The stages of ONE batch depend on each other (QSI:101-352 runs them in sequence), but a serving loop keeps batches in
 * flight — in production the host loads + decrypts batch t's candidates (PIS:717-724, AesGcmCryptoService.java:126-166)
 * while the GPU already routes batch t+1 — and stages of DIFFERENT batches are independent.  fspann_tick_dev enqueues, as
 * ONE kernel whose workgroups each take one role,
 *     encode of one batch   (= fspann_encode_dev, exact fp64 coding),
 *     Route of another      (= fspann_route_dev with limit = cap = B, no counters: stage A + A.5, F_q),
 *     Refine of a third     (= fspann_refine_dev, or fspann_refine_store_dev when ref_cand_dev is NULL),
 * so that the latency-bound Route workgroups run under the bandwidth-bound Refine ones.  Each part is optional (nq = 0)
 * and produces exactly what its stand-alone call produces; dependencies between the parts of one batch are the caller's
 * (stream order between calls: encode in tick t, route in t+1, refine in t+2 or later).  When a part does not qualify for
 * the shared kernel (fp64 inputs, B > 256, bounded select not legal: see fspann_set_route_mode) the call falls back to the
 * stand-alone kernels in stream order — same results.
 * Hand-over buffer: a query the bounded select cannot hold is finished by the full select.  With route_handover_dev = NULL
 * that happens in a second small launch inside this call.  With a buffer of fspann_route_handover_bytes() that travels
 * with the batch (same pointer as ref_handover_dev when that batch is refined, together with its codes), the workgroup
 * that refines the query finishes its Route first: no extra launch.                                                       */
typedef struct fspann_tick {
    /* encode role */
    int64_t nq_encode;
    const void* enc_q_dev;            /* [nq_encode][dim] */
    int32_t enc_dtype, pad0;
    uint64_t* enc_codes_dev;          /* [nq_encode][T*D][W] */
    int32_t* enc_bad_dev;             /* [nq_encode], 1 where NaN/Inf (may be NULL) */
    /* route role */
    int64_t nq_route;
    const uint64_t* route_codes_dev;
    int32_t route_probe_override, route_limit;   /* route_limit = B of the Refine that will consume F_q (also its row pitch) */
    int32_t* route_ids_dev;           /* [nq_route][route_limit] */
    int32_t* route_count_dev;         /* [nq_route] */
    void* route_handover_dev;         /* fspann_route_handover_bytes(), or NULL */
    /* refine role */
    int64_t nq_refine;
    const void* ref_q_dev;
    int32_t ref_q_dtype, ref_cand_dtype;
    const void* ref_cand_dev;         /* [nq_refine][B][dim] decrypted rows, or NULL: rows read from the resident store by id */
    int64_t ref_B;
    int32_t* ref_ids_dev;             /* F_q of that batch, [nq_refine][B] (written when a PENDING query is finished here) */
    int32_t* ref_count_dev;           /* [nq_refine] */
    const uint64_t* ref_codes_dev;    /* codes of that batch + the hand-over buffer its Route wrote (both NULL: none PENDING) */
    void* ref_handover_dev;
    int32_t ref_probe_override, k;
    int32_t* out_ids_dev;             /* [nq_refine][k] */
    double* out_dist_dev;
    int32_t* out_count_dev;
    int32_t* scored_dev;              /* may be NULL */
} fspann_tick;
size_t fspann_route_handover_bytes(fspann_ctx* ctx, int64_t nq, int probe_override);
int fspann_tick_dev(fspann_ctx* ctx, const fspann_tick* t);
/* 1 if the last fspann_tick_dev ran as one shared kernel, 0 if it fell back to the stand-alone kernels. */
int fspann_last_tick_fused(fspann_ctx* ctx);

/* ---- exact ground truth + evaluation metrics --------------------------------------------------------
 * GroundtruthPrecompute.run (api/.../GroundtruthPrecompute.java:218-272): per query the k base vectors with the smallest
 * squared L2 distance, ties by LOWER id (:167-171), ascending; the distance arithmetic is the reference's (float
 * subtraction, fp64 squares summed in dimension order, :142-163), so the ids equal a JVM run's.  base [n][dim], q [nq][dim]
 * fp32 in device memory (fvecs data); out_ids [nq][k] (-1 beyond n), out_d2 [nq][k] squared distances (may be NULL).
 * fspann_eval_metrics_dev = ForwardSecureANNSystem.computeMetricsAtK (FSA:770-835): recall@k and distance ratio@k per
 * query (ratio NaN when the reference yields NaN); ann ids [nq][ann_stride] with ann_count (NULL: all), gt [nq][gt_stride].  */
int fspann_groundtruth_dev(fspann_ctx* ctx, int64_t n, const float* base_dev, int64_t nq, const float* q_dev, int dim, int k,
                           int32_t* out_ids_dev, double* out_d2_dev);
int fspann_eval_metrics_dev(fspann_ctx* ctx, int64_t n, const float* base_dev, int64_t nq, const float* q_dev, int dim, int k,
                            const int32_t* ann_ids_dev, int64_t ann_stride, const int32_t* ann_count_dev, const int32_t* gt_ids_dev,
                            int64_t gt_stride, double* recall_dev, double* ratio_dev);

/* ---- host candidate pipeline -------------------------------------------------------------------
 * QSI stage B's host half at batch scale: for every id of F_q the reference does loadPointIfActive (one RocksDB get + one
 * Java-deserialised .point file, PIS:717-724, common/RocksDBMetadataManager.java:530-544) and decryptFromPoint
 * (Cipher.getInstance + AES-256-GCM open + big-endian fp64 decode, crypto/AesGcmCryptoService.java:126-166,261-277) —
 * 89-93 % of its latency.  Here: ONE packed in-memory point store (record = key version, 12-byte IV, 8*dim-byte big-endian
 * fp64 ciphertext || 16-byte tag), AES-GCM opened by a thread pool straight into pinned [nq][B][dim] staging, and a
 * three-stage pipeline (Route of batch i+1 | decrypt of batch i | H2D + Refine of batch i-1).  Crypto is the reference's, bit
 * for bit: AAD "id:%s|v:%d|d:%d" with the decimal id (common/EncryptedPoint.java:80-83), K_v = HMAC-SHA256(K_M, int32_be(v))
 * (keymanagement/KeyManager.java:221-237), Migrate = open with the record's version, seal with the current one under a
 * fresh IV (keymanagement/KeyRotationServiceImpl.java:215-289) — records sealed by a JVM can be imported and read, and vice
 * versa.  A JVM deployment keeps its own decrypt loop and hands rows to fspann_refine[_dev]; this is the native alternative.
 * libcrypto (OpenSSL 3) is bound at run time ($FSPANN_CRYPTO_LIB, else libcrypto.so.3).  Decrypt stays on the HOST.        */
typedef struct fspann_pointstore fspann_pointstore;
typedef struct fspann_pipeline fspann_pipeline;
int fspann_pointstore_create(int64_t n, int dim, fspann_pointstore** out);
void fspann_pointstore_destroy(fspann_pointstore* ps);
int fspann_pointstore_set_master_key(fspann_pointstore* ps, const uint8_t* key32);
int fspann_pointstore_current_version(fspann_pointstore* ps);
int fspann_pointstore_rotate(fspann_pointstore* ps, int* new_version);              /* KeyRotationServiceImpl.rotateKeyOnly :292-305 */
int fspann_pointstore_retire(fspann_pointstore* ps, int version);                   /* KeyManager retire :274-317 */
/* encryptToPoint (AesGcmCryptoService.java:55-112) of handles [h0, h0 + cnt): row i of `vectors` = handle h0 + i. */
int fspann_pointstore_encrypt(fspann_pointstore* ps, int64_t h0, int64_t cnt, const void* vectors, int dtype, int threads);
int fspann_pointstore_delete(fspann_pointstore* ps, int64_t h);
/* reencryptTouched: records older than the current version move to it; *reencrypted = how many did. */
int fspann_pointstore_reencrypt(fspann_pointstore* ps, const int32_t* handles, int64_t cnt, int threads, int64_t* reencrypted);
/* Stage B for a batch: ids [nq][B], count [nq] (negative = 0) -> dst [nq][B][dim] (dst_dtype), rows that fail to load or
 * open are skipped and the survivors packed to the front in F_q order (QSI:240-270); out_ids [nq][B], out_count [nq].   */
int fspann_pointstore_open_batch(fspann_pointstore* ps, int64_t nq, int64_t B, const int32_t* ids, const int32_t* count, void* dst, int dst_dtype,
                                 int32_t* out_ids, int32_t* out_count, int threads);
int fspann_pointstore_get_record(fspann_pointstore* ps, int64_t h, int32_t* version, uint8_t* iv12, uint8_t* ct /* 8*dim + 16 */);
int fspann_pointstore_put_record(fspann_pointstore* ps, int64_t h, int32_t version, const uint8_t* iv12, const uint8_t* ct);
int fspann_pointstore_stats(fspann_pointstore* ps, int64_t* opened, int64_t* failed);
/* The pipeline over one (finalized) context: submit host fp32 query batches, collect results in submission order.
 * Four batches may be in flight; submit blocks when all slots are taken, collect blocks until the oldest batch is done. */
int fspann_pipeline_create(fspann_ctx* ctx, fspann_pointstore* ps, int64_t nq_max, int64_t B, int k, int host_threads, fspann_pipeline** out);
int fspann_pipeline_submit(fspann_pipeline* p, int64_t nq, const float* q_host, uint64_t* ticket);
int fspann_pipeline_collect(fspann_pipeline* p, uint64_t* ticket, int64_t* nq, int32_t* out_ids, double* out_dist, int32_t* out_count);
int fspann_pipeline_stats(fspann_pipeline* p, double* route_ms, double* decrypt_ms, double* refine_ms, int64_t* batches);
void fspann_pipeline_destroy(fspann_pipeline* p);

/* ---- multi-GPU merge ---------------------------------------------------------
 * The reference is a single JVM with a serial query loop (ForwardSecureANNSystem.java:636): it has no collective.  Here
 * queries shard over GPUs (one context per GPU, index replicated, contiguous equal shards of the batch) and the ONLY
 * exchange is one RCCL all-gather of every rank's packed top-k, enqueued on the context's stream behind Refine.
 *   packed top-k of nq queries = [nq*k] int32 ids, padded to 8 bytes, then [nq*k] fp64 distances: hand
 *   fspann_refine*_dev `base` as out_ids and `base + fspann_topk_dist_offset()` as out_dist and no packing is needed.
 * Bootstrap: rank 0 calls fspann_comm_unique_id and ships the 128 bytes to the other ranks over the caller's own
 * transport; every rank then calls fspann_comm_create (collective: blocks until all `world` ranks have called).
 * librccl is bound at run time: $FSPANN_RCCL_LIB, an instance already in the process, else librccl.so.1.          */
typedef struct fspann_comm fspann_comm;
#define FSPANN_UNIQUE_ID_BYTES 128
size_t fspann_topk_bytes(int64_t nq, int k);
size_t fspann_topk_dist_offset(int64_t nq, int k);
int fspann_comm_available(void);   /* 1 if librccl could be bound in this process (lets all ranks agree before the collective create) */
int fspann_comm_unique_id(void* id_out /* FSPANN_UNIQUE_ID_BYTES */);
/* The communicator launches on ctx's stream and keeps ctx alive: an fspann_ctx_destroy(ctx) that comes first is finished by the
 * last fspann_comm_destroy. */
int fspann_comm_create(fspann_ctx* ctx, const void* unique_id, int world, int rank, fspann_comm** out);
int fspann_comm_destroy(fspann_comm* comm);
int fspann_comm_info(fspann_comm* comm, int* world, int* rank, const char** library);
/* gathered_dev = world * fspann_topk_bytes(nq_local, k) bytes, rank r's block at r * fspann_topk_bytes(). */
int fspann_allgather_topk_dev(fspann_comm* comm, int64_t nq_local, int k, const void* local_packed_dev, void* gathered_dev);

/* Measurement aid (bench.py roofline.peak_measured): GB/s at which this device streams `bytes` of HBM through a pure
 * 16-byte-load kernel (best of `reps`, and of the default and the nt cache policy); pick bytes well above the 256 MiB Infinity Cache. */
int fspann_hbm_read_peak(fspann_ctx* ctx, size_t bytes, int reps, double* gb_per_s);
/* The same kernel on a launch of the hot path's size: `reps` launches, each reading the next `window` bytes of a `bytes`
 * buffer (so every launch reads cold HBM), average rate over the launches.  What a short launch can reach at all: the
 * ramp at the start and the drain at the end of a launch are not amortised over 134 MB as they are over 4 GiB.     */
int fspann_hbm_read_window(fspann_ctx* ctx, size_t bytes, size_t window, int reps, double* gb_per_s);

/* ---- device memory helpers (so non-torch callers can own HBM buffers) ------------------- */
int fspann_dev_alloc(fspann_ctx* ctx, size_t bytes, void** out);
int fspann_dev_free(fspann_ctx* ctx, void* p);
int fspann_h2d(fspann_ctx* ctx, void* dst_dev, const void* src, size_t bytes);
int fspann_d2h(fspann_ctx* ctx, void* dst, const void* src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* FSPANN_H */
