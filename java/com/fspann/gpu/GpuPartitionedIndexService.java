package com.fspann.gpu;

import com.fspann.common.EncryptedPoint;
import com.fspann.common.EncryptedPointBuffer;
import com.fspann.common.IndexService;
import com.fspann.common.QueryToken;
import com.fspann.config.SystemConfig;
import com.fspann.index.paper.Coding;
import com.fspann.index.paper.GFunctionRegistry;
import com.fspann.index.paper.PartitionedIndexService;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayList;
import java.util.BitSet;
import java.util.HashMap;
import java.util.HashSet;
import java.util.List;
import java.util.Map;
import java.util.Set;

/**
 * Drop-in for {@link PartitionedIndexService} with Route on the GPU.
 *
 * The reference class is {@code final} and wired by constructor (ForwardSecureANNSystem.java:316-322), so this is a
 * COMPOSITION: everything that stays on the host in the reference — encrypt + persist on insert (PIS:265-347),
 * loadPointIfActive (PIS:717-724), the point buffer — is forwarded to a stock PartitionedIndexService; the routing
 * state (registry, partitions) is built a second time in HBM from the same inputs in the same order, and the two Route
 * operators ({@link #lookupCandidatesWithScores}, {@link #lookupCandidateIds}) plus the probe override / last-touched /
 * raw-count accessors are served from there.  Wiring edit in ForwardSecureANNSystem: construct this instead of
 * PartitionedIndexService (same four arguments + a device index) and give it to {@link GpuQueryServiceImpl}.
 *
 * Order of {@code staged} (which HashMap iteration order, and therefore partition membership among equal keys, depends
 * on): ids 0..998 are parked until the 1000th insert initialises the registry, so the staged order is 999, 1000, ..., n-1,
 * 0, ..., 998 (PIS:280-312, 821-831) — exactly what fspann_build_index assumes when {@code order == null}.
 *
 * Not compiled in the build container (no JDK).  Its Python twin, fspann-query-system_amd/operators.py
 * (class PartitionedIndexService), IS exercised by the test-suite against the same C ABI.
 */
public final class GpuPartitionedIndexService implements IndexService, AutoCloseable {
    private final PartitionedIndexService host;          // unchanged host-side behaviour (crypto, RocksDB, .point files)
    private final SystemConfig cfg;
    private final int device;
    private long ctx = 0;
    private int dim = -1, tables, divisions, m, lambda, words;
    private final List<String> idOf = new ArrayList<>();
    private final Map<String, Integer> handleOf = new HashMap<>();
    // Plaintext rows never pile up in the JVM heap: they go to a fixed direct buffer (CHUNK_BYTES) that is handed to
    // fspann_build_append whenever it is full — the library codes each piece on arrival and keeps only the codes in HBM.
    // Until the registry exists (the first MIN_SAMPLE_SIZE - 1 inserts, PIS:280-312) the rows are parked here, as the
    // reference parks them in pendingVectors.
    private static final long CHUNK_BYTES = 64L << 20;
    private final List<double[]> parked = new ArrayList<>();
    private ByteBuffer chunk;
    private int chunkRows = 0, chunkCap = 0;
    private long appended = 0;
    private volatile boolean frozen = false;
    private final ThreadLocal<Integer> probeOverride = ThreadLocal.withInitial(() -> -1);      // PIS:68-73
    private final ThreadLocal<Set<String>> lastTouched = ThreadLocal.withInitial(HashSet::new);
    private volatile int lastRawVisited = 0;                                                      // PIS:63 (shared, as in the reference)
    private volatile boolean countersWanted = true;     // lastRawVisited / lastCandKept are profiler fields: see setCountersWanted

    public GpuPartitionedIndexService(PartitionedIndexService host, SystemConfig cfg, int device) {
        this.host = java.util.Objects.requireNonNull(host, "host");
        this.cfg = java.util.Objects.requireNonNull(cfg, "cfg");
        this.device = device;
    }

    /** Direct buffer in native order; a size that does not fit a Java buffer is a caller bug, not something to truncate. */
    static ByteBuffer buf(long bytes) {
        if (bytes < 0 || bytes > Integer.MAX_VALUE)
            throw new IllegalArgumentException("direct buffer of " + bytes + " bytes: hand the data over in pieces");
        return ByteBuffer.allocateDirect((int) bytes).order(ByteOrder.nativeOrder());
    }

    /** Native context + the JVM's own GFunctions, as soon as GFunctionRegistry is initialised. */
    private void openNative() {
        SystemConfig.PaperConfig pc = cfg.getPaper();
        tables = pc.getTables(); divisions = pc.getDivisions(); m = pc.getM(); lambda = pc.getLambda();
        words = (m * lambda + 63) / 64;
        SystemConfig.RuntimeConfig rc = cfg.getRuntime();
        long[] h = new long[1];
        FspannNative.check(FspannNative.ctxCreate(device, new int[]{tables, divisions, m, lambda, dim, 64, 5, rc.getProbeOverride(),
                rc.getMaxGlobalCandidates(), rc.getRefinementLimit(), rc.getHammingPrefilterThreshold(), 0}, h));
        ctx = h[0];
        // the JVM's own alpha / r / omega, bit for bit (Math.log / Math.cos are not portable across runtimes)
        int TD = tables * divisions;
        ByteBuffer a = buf(8L * TD * m * dim), r = buf(8L * TD * m), w = buf(8L * TD * m);
        for (int t = 0; t < tables; t++)
            for (int d = 0; d < divisions; d++) {
                Coding.GFunction g = GFunctionRegistry.get(dim, t, d);
                for (int j = 0; j < m; j++) {
                    for (int i = 0; i < dim; i++) a.putDouble(g.alpha[j][i]);
                    r.putDouble(g.r[j]);
                    w.putDouble(g.omega[j]);
                }
            }
        FspannNative.check(FspannNative.setGfunctions(ctx, a, r, w));
        chunkCap = (int) Math.max(1, Math.min(CHUNK_BYTES / (8L * dim), 1 << 20));
        chunk = buf(8L * chunkCap * dim);
        FspannNative.check(FspannNative.buildBegin(ctx, Math.max(chunkCap, 1 << 20)));   // a hint: the code buffer grows with the rows
        for (double[] v : parked) appendRow(v);            // handles are insertion numbers: rows go over in insertion order
        parked.clear();
    }

    private void appendRow(double[] v) {
        for (double x : v) chunk.putDouble(x);
        if (++chunkRows == chunkCap) flushChunk();
    }

    private void flushChunk() {
        if (chunkRows == 0) return;
        FspannNative.check(FspannNative.buildAppend(ctx, chunkRows, chunk, FspannNative.F64));
        appended += chunkRows;
        chunkRows = 0;
        chunk.clear();
    }

    // ---- IndexService ------------------------------------------------------------------------------------------------
    @Override public void insert(String id, double[] vector) {
        host.insert(id, vector);                          // validation, sample buffer, encrypt, persist: PIS:265-347
        synchronized (this) {
            if (frozen) throw new IllegalStateException("Index already finalized");
            if (dim < 0) dim = vector.length;
            handleOf.put(id, idOf.size());
            idOf.add(id);
            if (ctx == 0 && GFunctionRegistry.isInitialized()) openNative();
            if (ctx != 0) appendRow(vector); else parked.add(vector.clone());
        }
    }

    @Override public synchronized void finalizeForSearch() {
        host.finalizeForSearch();                         // flushes the parked points, initialises GFunctionRegistry (PIS:789-845)
        if (frozen) return;
        if (ctx == 0) openNative();                       // fewer than MIN_SAMPLE_SIZE inserts: the registry exists only now
        flushChunk();
        int n = idOf.size();
        ByteBuffer jh = buf(4L * n);
        for (String id : idOf) jh.putInt(id.hashCode());           // decides HashMap iteration order (DESIGN.md "Java order key")
        FspannNative.check(FspannNative.setIdMeta(ctx, n, jh, null));
        // GreedyPartitioner.build on the GPU over the coded rows, staged order = the reference's (null).  A staging map that
        // treeifies a bin is ordered by the library's JDK model; only equal hashCodes of non-decimal ids inside such a bin make
        // it throw IllegalStateException (String.compareTo unknown to it): fall back to the host class then.
        FspannNative.check(FspannNative.buildFinish(ctx, null));
        chunk = null;
        frozen = true;
    }

    /** Mirror of metadata.isDeleted(id) (PIS:739): call beside the metadata update that marks / unmarks a point as deleted.
     *  Takes effect for the next query, on this service and on every clone of its native context; nothing is un-frozen. */
    public void markDeleted(String id, boolean deleted) {
        Integer h = handleOf.get(id);
        if (h == null || ctx == 0) return;
        ByteBuffer hb = buf(4);
        hb.putInt(0, h);
        FspannNative.check(FspannNative.setDeleted(ctx, hb, 1, deleted ? 1 : 0));
    }

    /** lastCandKept / getLastRawCandidateCount are profiler fields of the reference (QSI:417-474).  Producing them forces the
     *  full select over every probed partition (2.5 x the bounded select): ask for them only while a profiler reads them. */
    public void setCountersWanted(boolean wanted) { countersWanted = wanted; }
    boolean countersWanted() { return countersWanted; }

    @Override public void updateCachedPoint(EncryptedPoint point) { host.updateCachedPoint(point); }
    @Override public EncryptedPointBuffer getPointBuffer() { return host.getPointBuffer(); }

    // ---- Route (PIS:459-715) -------------------------------------------------------------------------------------------
    /** Same record as PartitionedIndexService.CandidateWithScore (PIS:82-89). */
    public record CandidateWithScore(String id, long hammingDist) implements Comparable<CandidateWithScore> {
        @Override public int compareTo(CandidateWithScore o) { return Long.compare(hammingDist, o.hammingDist); }
    }

    private ByteBuffer tokenCodes(QueryToken token) {
        if (!frozen) throw new IllegalStateException("Index not finalized");                       // PIS:594
        if (token.getDimension() != dim) return null;                                              // unknown dimension -> empty (PIS:598)
        BitSet[][] codes = token.getBitCodes();
        if (codes == null) throw new IllegalStateException("MSANNP violation: QueryToken missing BitSet codes");   // PIS:602
        if (codes.length != tables) throw new IllegalStateException("Token tables mismatch");      // PIS:605
        ByteBuffer c = buf(8L * tables * divisions * words);
        for (int t = 0; t < tables; t++) {
            if (codes[t].length != divisions) throw new IllegalStateException("Token divisions mismatch");   // PIS:630
            for (int d = 0; d < divisions; d++) {
                long[] w = codes[t][d].toLongArray();
                for (int k = 0; k < words; k++) c.putLong(k < w.length ? w[k] : 0L);
            }
        }
        return c;
    }

    /** limit = Integer.MAX_VALUE: lookupCandidatesWithScores; limit = HARD_CAP: lookupCandidateIds; limit = B: stage A.5. */
    List<CandidateWithScore> route(QueryToken token, int limit, boolean withCounters, int[] keptOut) {
        ByteBuffer c = tokenCodes(token);
        List<CandidateWithScore> out = new ArrayList<>();
        if (c == null) return out;
        int po = probeOverride.get();
        long cap = Math.max(1, Math.min((long) limit, FspannNative.routeMaxCandidates(ctx, po)));
        ByteBuffer ids = buf(4 * cap), sc = buf(4 * cap), cnt = buf(4);
        ByteBuffer kept = withCounters ? buf(4) : null, raw = withCounters ? buf(4) : null;
        // a query whose bestScore map treeifies a bin is finished by the library's JDK model inside this call; IllegalStateException
        // only for equal hashCodes of non-decimal ids inside such a bin (String.compareTo unknown to the library)
        FspannNative.check(FspannNative.route(ctx, 1, c, po, limit, cap, ids, sc, cnt, kept, raw));
        int n = cnt.getInt(0);
        Set<String> touched = lastTouched.get();
        touched.clear();
        for (int i = 0; i < n; i++) {
            String id = idOf.get(ids.getInt(4 * i));
            out.add(new CandidateWithScore(id, sc.getInt(4 * i)));
            touched.add(id);
        }
        if (withCounters) { lastRawVisited = raw.getInt(0); if (keptOut != null) keptOut[0] = kept.getInt(0); }
        return out;
    }

    /**
     * Stage A + A.5 for a BATCH of tokens with ONE fspann_route (GpuQueryServiceImpl.searchBatch): entry q of the result is what
     * route(tokens.get(q), limit, ...) returns; a token of another dimension yields an empty list (PIS:598).  keptOut / rawOut
     * (length = tokens.size(), may be null) receive lastCandKept / rawSeen per token when withCounters is set.  probes: the probe
     * override of this pass (-1 = default; the caller's adaptive retry passes 10).
     */
    List<List<CandidateWithScore>> routeBatch(List<QueryToken> tokens, int limit, int probes, boolean withCounters, int[] keptOut, int[] rawOut) {
        final int nq = tokens.size();
        List<List<CandidateWithScore>> out = new ArrayList<>(nq);
        final long codeBytes = 8L * tables * divisions * words;
        ByteBuffer codes = buf(codeBytes * nq);
        int[] slot = new int[nq];
        int live = 0;
        for (int q = 0; q < nq; q++) {
            out.add(new ArrayList<>());
            ByteBuffer c = tokenCodes(tokens.get(q));
            slot[q] = -1;
            if (c == null) continue;
            c.flip();
            codes.put(c);
            slot[q] = live++;
        }
        if (live == 0) return out;
        long cap = Math.max(1, Math.min((long) limit, FspannNative.routeMaxCandidates(ctx, probes)));
        ByteBuffer ids = buf(4 * cap * live), sc = buf(4 * cap * live), cnt = buf(4L * live);
        ByteBuffer kept = withCounters ? buf(4L * live) : null, raw = withCounters ? buf(4L * live) : null;
        FspannNative.check(FspannNative.route(ctx, live, codes, probes, limit, cap, ids, sc, cnt, kept, raw));
        Set<String> touched = lastTouched.get();
        touched.clear();
        for (int q = 0; q < nq; q++) {
            if (slot[q] < 0) continue;
            final int s = slot[q], n = cnt.getInt(4 * s);
            List<CandidateWithScore> lq = out.get(q);
            for (int i = 0; i < n; i++) {
                String id = idOf.get(ids.getInt((int) (4 * (s * cap + i))));
                lq.add(new CandidateWithScore(id, sc.getInt((int) (4 * (s * cap + i)))));
                if (q == nq - 1) touched.add(id);                                   // getLastTouchedIds describes the last token
            }
            if (withCounters) {
                if (keptOut != null) keptOut[q] = kept.getInt(4 * s);
                if (rawOut != null) rawOut[q] = raw.getInt(4 * s);
                if (q == nq - 1) lastRawVisited = raw.getInt(4 * s);
            }
        }
        return out;
    }

    public List<CandidateWithScore> lookupCandidatesWithScores(QueryToken token) {
        return route(token, Integer.MAX_VALUE, true, null);
    }

    public List<String> lookupCandidateIds(QueryToken token) {
        int hardCap = Math.max(cfg.getRuntime().getMaxGlobalCandidates(), cfg.getRuntime().getRefinementLimit());   // PIS:474-477, truncation :558-565
        List<String> out = new ArrayList<>();
        for (CandidateWithScore cs : route(token, hardCap, true, null)) out.add(cs.id());
        return out;
    }

    public EncryptedPoint loadPointIfActive(String id) { return host.loadPointIfActive(id); }     // stays on the host (PIS:717-724)
    public int numTables() { return tables; }
    public boolean isFrozen() { return frozen; }
    public Set<String> getLastTouchedIds() { return new HashSet<>(lastTouched.get()); }
    public int getLastTouchedCount() { return lastTouched.get().size(); }
    public void setProbeOverride(int probes) { probeOverride.set(probes); }                         // PIS:868
    public void clearProbeOverride() { probeOverride.set(-1); }                                     // PIS:872
    public int getLastRawCandidateCount() { return lastRawVisited; }
    public int getDefaultMaxProbes() { return FspannNative.effectiveProbes(ctx, -1); }              // PIS:880-894
    long nativeContext() { return ctx; }
    int dimension() { return dim; }
    int handleOf(String id) { Integer h = handleOf.get(id); return h == null ? -1 : h; }

    @Override public synchronized void close() {
        if (ctx != 0) { FspannNative.ctxDestroy(ctx); ctx = 0; }
    }
}
