package com.fspann.gpu;

import com.fspann.common.EncryptedPoint;
import com.fspann.common.KeyLifeCycleService;
import com.fspann.common.KeyVersion;
import com.fspann.common.QueryResult;
import com.fspann.common.QueryToken;
import com.fspann.config.SystemConfig;
import com.fspann.crypto.CryptoService;
import com.fspann.crypto.ReencryptionTracker;
import com.fspann.query.core.QueryTokenFactory;
import com.fspann.query.service.QueryService;

import java.nio.ByteBuffer;
import java.util.ArrayList;
import java.util.Collections;
import java.util.HashSet;
import java.util.List;
import java.util.Set;

/**
 * Drop-in for QueryServiceImpl (query/src/main/java/com/fspann/query/service/QueryServiceImpl.java:101-352): the same
 * control flow — decrypt the query, Route, keep the first B (stage A.5), load + decrypt each candidate ON THE HOST
 * (QSI:238-271, unchanged: loadPointIfActive + CryptoService.decryptFromPoint with the point's own key version), score,
 * stable top-K, one adaptive retry with 10 probes — with Route and the distance / top-K part of Refine on the GPU.
 * The decrypted rows are packed into one direct buffer and handed to fspann_refine; nothing encrypted or keyed ever
 * reaches the device.  Metric getters mirror QSI:417-474.  Not re-entrant per instance, like the reference (QSI:45-64).
 *
 * Not compiled in the build container (no JDK); Python twin: operators.py (class QueryServiceImpl), tested.
 */
public final class GpuQueryServiceImpl implements QueryService {
    private final GpuPartitionedIndexService index;
    private final CryptoService cryptoService;
    private final KeyLifeCycleService keyService;
    private final QueryTokenFactory tokenFactory;
    private final SystemConfig cfg;
    private ReencryptionTracker reencTracker;
    private final ThreadLocal<Integer> refinementLimitOverride = ThreadLocal.withInitial(() -> -1);   // QSI:454-466
    private final Set<String> touchedThisSession = new HashSet<>();
    private volatile long lastServerNs, lastClientNs, lastDecryptNs;
    private volatile int lastCandTotal, lastCandKept, lastCandDecrypted, lastReturned, lastUniqueCandidates, lastEffectiveLimit;
    private volatile List<String> lastFinalIds = Collections.emptyList();

    public GpuQueryServiceImpl(GpuPartitionedIndexService index, CryptoService cryptoService, KeyLifeCycleService keyService,
                               QueryTokenFactory tf, SystemConfig cfg) {
        this.index = java.util.Objects.requireNonNull(index, "index");
        this.cryptoService = java.util.Objects.requireNonNull(cryptoService, "cryptoService");
        this.keyService = java.util.Objects.requireNonNull(keyService, "keyService");
        this.tokenFactory = tf;
        this.cfg = java.util.Objects.requireNonNull(cfg, "cfg");
    }

    private static ByteBuffer buf(long bytes) { return GpuPartitionedIndexService.buf(bytes); }   // refuses sizes beyond a Java buffer
    /** The native context's PINNED block as a direct buffer (fspann_host_buffer): the decrypted rows are packed straight into it and
     *  reach the GPU by one DMA; a JVM direct buffer is pageable memory and goes through the runtime's staging copies. */
    private ByteBuffer pinned(long bytes) {
        if (bytes < 0 || bytes > Integer.MAX_VALUE) throw new IllegalArgumentException("pinned buffer of " + bytes + " bytes: hand the data over in pieces");
        ByteBuffer b = FspannNative.hostBuffer(index.nativeContext(), Math.max(bytes, 16));
        if (b == null) throw new OutOfMemoryError(FspannNative.lastError());
        return b.order(java.nio.ByteOrder.nativeOrder());
    }

    private static boolean isValid(double[] v) {                                // QSI:407-413
        if (v == null) return false;
        for (double x : v) if (Double.isNaN(x) || Double.isInfinite(x)) return false;
        return true;
    }

    @Override public List<QueryResult> search(QueryToken token) {
        if (token == null) return Collections.emptyList();                      // QSI:102-104
        touchedThisSession.clear();                                             // QSI:120
        final long t0 = System.nanoTime();
        long decryptNs = 0;
        final double[] q;
        try {                                                                    // QSI:124-135
            KeyVersion kv;
            try { kv = keyService.getVersion(token.getVersion()); } catch (RuntimeException e) { kv = keyService.getCurrentVersion(); }
            q = cryptoService.decryptQuery(token.getEncryptedQuery(), token.getIv(), kv.getKey());
        } catch (RuntimeException e) {
            return Collections.emptyList();
        }
        if (!isValid(q)) return Collections.emptyList();                         // QSI:137-140
        final int K = token.getTopK(), dim = q.length;
        boolean retried = false;
        try {
            while (true) {
                int limit = getEffectiveRefinementLimit(cfg.getRuntime().getRefinementLimit());
                lastEffectiveLimit = limit;
                int[] kept = new int[1];
                // stage A + A.5 on the GPU: the first `limit` entries of the reference's list (HashMap order, stable by score)
                // lastCandTotal / lastCandKept are profiler fields (QSI:417-474): requested only while index.setCountersWanted(true)
                // — they force the full select over every probed partition; without them the bounded select runs (-1 = not collected)
                final boolean counters = index.countersWanted();
                List<GpuPartitionedIndexService.CandidateWithScore> fq = index.route(token, limit, counters, kept);
                lastCandTotal = counters ? index.getLastRawCandidateCount() : -1;
                lastCandKept = counters ? kept[0] : -1;
                lastUniqueCandidates = fq.size();
                if (fq.isEmpty()) { lastReturned = 0; return Collections.emptyList(); }
                // stage B, host part — exactly the reference's loop (QSI:238-271): load, decrypt with the point's own version, validate
                final long td0 = System.nanoTime();
                ByteBuffer rows = pinned(8L * fq.size() * dim), ids = buf(4L * fq.size());
                List<String> rowId = new ArrayList<>(fq.size());
                for (GpuPartitionedIndexService.CandidateWithScore c : fq) {
                    try {
                        EncryptedPoint ep = index.loadPointIfActive(c.id());
                        if (ep == null) continue;
                        double[] v = cryptoService.decryptFromPoint(ep, keyService.getVersion(ep.getKeyVersion()).getKey());
                        if (!isValid(v) || v.length != dim) continue;
                        for (double x : v) rows.putDouble(x);
                        ids.putInt(rowId.size());
                        rowId.add(c.id());
                        touchedThisSession.add(c.id());
                    } catch (Exception e) {
                        // per-candidate failures are swallowed (QSI:264-269)
                    }
                }
                decryptNs += System.nanoTime() - td0;
                lastCandDecrypted = rowId.size();
                if (rowId.isEmpty()) { lastReturned = 0; return Collections.emptyList(); }
                // stage B distances + stage C on the GPU: sequential fp64 L2 (QSI:364-372), stable sort, first K (QSI:298-316)
                int B = rowId.size();
                ByteBuffer qb = buf(8L * dim), nb = buf(4), oi = buf(4L * K), od = buf(8L * K), oc = buf(4), sc = buf(4);
                for (double x : q) qb.putDouble(x);
                nb.putInt(B);
                FspannNative.check(FspannNative.refine(index.nativeContext(), 1, qb, rows, FspannNative.F64, B, ids, nb, K, oi, od, oc, sc));
                int eff = oc.getInt(0);
                List<QueryResult> out = new ArrayList<>(eff);
                List<String> fin = new ArrayList<>(eff);
                for (int i = 0; i < eff; i++) {
                    String id = rowId.get(oi.getInt(4 * i));
                    out.add(new QueryResult(id, od.getDouble(8 * i)));
                    fin.add(id);
                }
                lastReturned = eff;
                lastFinalIds = fin;
                if (!retried && (lastReturned < K || lastCandDecrypted < 10 * K)) {   // QSI:327-337,444-447
                    retried = true;
                    index.setProbeOverride(10);
                    continue;                                                        // touchedThisSession keeps both passes (QSI:120)
                }
                return out;
            }
        } finally {                                                              // QSI:342-351
            index.clearProbeOverride();
            lastServerNs = System.nanoTime() - t0;
            lastDecryptNs = decryptNs;
            lastClientNs = 0;
            if (reencTracker != null && !touchedThisSession.isEmpty()) reencTracker.record(new HashSet<>(touchedThisSession));
        }
    }

    /**
     * The batched mirror of search(): the reference's runQueries loop calls search(token) once per query
     * (ForwardSecureANNSystem.java:636-748), which costs a GPU round trip per query per stage (bench.py operator_surface.per_query).
     * Here the tokens of a batch share ONE fspann_route, the host load + decrypt loop (QSI:238-271, unchanged) runs over every
     * F_q into one direct buffer, ONE fspann_refine scores all of them, and the adaptive retry (QSI:327-337) reruns — again as
     * one batch — exactly the queries that came back short.  Entry q is what search(tokens.get(q)) returns; the metric getters
     * describe the last token; the re-encryption tracker records every token's touched ids.  All tokens must ask for the same
     * topK (fspann_refine takes one k per call); otherwise the tokens are searched one by one.
     * Python twin: operators.QueryServiceImpl.searchBatch (tests/test_gpu_golden.py).
     */
    public List<List<QueryResult>> searchBatch(List<QueryToken> tokens) {
        final int nq = tokens.size();
        List<List<QueryResult>> results = new ArrayList<>(nq);
        for (int i = 0; i < nq; i++) results.add(Collections.emptyList());
        int K = -1;
        boolean sameK = true;
        for (QueryToken t : tokens) if (t != null) { if (K < 0) K = t.getTopK(); else sameK &= (K == t.getTopK()); }
        if (K < 0) return results;
        if (!sameK) { for (int i = 0; i < nq; i++) results.set(i, search(tokens.get(i))); return results; }
        final long t0 = System.nanoTime();
        long decryptNs = 0;
        double[][] qv = new double[nq][];
        List<Integer> active = new ArrayList<>();
        for (int i = 0; i < nq; i++) {                                            // QSI:102-140 per token
            QueryToken token = tokens.get(i);
            if (token == null) continue;
            try {
                KeyVersion kv;
                try { kv = keyService.getVersion(token.getVersion()); } catch (RuntimeException e) { kv = keyService.getCurrentVersion(); }
                double[] q = cryptoService.decryptQuery(token.getEncryptedQuery(), token.getIv(), kv.getKey());
                if (isValid(q)) { qv[i] = q; active.add(i); }
            } catch (RuntimeException e) { /* empty result */ }
        }
        final int limit = getEffectiveRefinementLimit(cfg.getRuntime().getRefinementLimit());
        lastEffectiveLimit = limit;
        final boolean counters = index.countersWanted();
        int[] kept = new int[nq], raw = new int[nq], decrypted = new int[nq], returned = new int[nq], unique = new int[nq];
        List<Set<String>> touched = new ArrayList<>(nq);
        for (int i = 0; i < nq; i++) touched.add(new HashSet<>());
        List<String> finLast = Collections.emptyList();
        try {
            int probes = -1;
            for (int attempt = 0; attempt < 2 && !active.isEmpty(); attempt++) {
                List<QueryToken> toks = new ArrayList<>(active.size());
                for (int i : active) toks.add(tokens.get(i));
                int[] keptA = new int[active.size()], rawA = new int[active.size()];
                List<List<GpuPartitionedIndexService.CandidateWithScore>> fqs = index.routeBatch(toks, limit, probes, counters, keptA, rawA);   // stage A + A.5: one call
                // stage B, host part — the reference's loop (QSI:238-271) over every F_q; rows packed as [query][Bmax][dim]
                final long td0 = System.nanoTime();
                int bmax = 0;
                for (List<GpuPartitionedIndexService.CandidateWithScore> fq : fqs) bmax = Math.max(bmax, fq.size());
                if (bmax == 0) break;
                final int dim = qv[active.get(0)].length, na = active.size();
                ByteBuffer rows = pinned(8L * na * bmax * dim), ids = buf(4L * na * bmax), cnt = buf(4L * na), qb = buf(8L * na * dim);
                List<List<String>> rowIds = new ArrayList<>(na);
                for (int a = 0; a < na; a++) {
                    final int i = active.get(a);
                    kept[i] = counters ? keptA[a] : -1; raw[i] = counters ? rawA[a] : -1; unique[i] = fqs.get(a).size();
                    List<String> rid = new ArrayList<>(fqs.get(a).size());
                    int pos = (int) (8L * a * bmax * dim);
                    for (GpuPartitionedIndexService.CandidateWithScore c : fqs.get(a)) {
                        try {
                            EncryptedPoint ep = index.loadPointIfActive(c.id());
                            if (ep == null) continue;
                            double[] v = cryptoService.decryptFromPoint(ep, keyService.getVersion(ep.getKeyVersion()).getKey());
                            if (!isValid(v) || v.length != dim) continue;
                            for (double x : v) { rows.putDouble(pos, x); pos += 8; }
                            ids.putInt(4 * (a * bmax + rid.size()), rid.size());
                            rid.add(c.id());
                            touched.get(i).add(c.id());
                        } catch (Exception e) { /* per-candidate failures are swallowed (QSI:264-269) */ }
                    }
                    rowIds.add(rid);
                    decrypted[i] = rid.size();
                    cnt.putInt(4 * a, rid.size());
                    for (int j = 0; j < dim; j++) qb.putDouble(8 * (a * dim + j), qv[i][j]);
                    if (rid.isEmpty()) { results.set(i, Collections.emptyList()); returned[i] = 0; }
                }
                decryptNs += System.nanoTime() - td0;
                // stage B distances + stage C: one call for the batch (a query without rows has count 0 and gets an empty result)
                ByteBuffer oi = buf(4L * na * K), od = buf(8L * na * K), oc = buf(4L * na), sc = buf(4L * na);
                FspannNative.check(FspannNative.refine(index.nativeContext(), na, qb, rows, FspannNative.F64, bmax, ids, cnt, K, oi, od, oc, sc));
                List<Integer> again = new ArrayList<>();
                for (int a = 0; a < na; a++) {
                    final int i = active.get(a);
                    if (rowIds.get(a).isEmpty()) continue;
                    int eff = oc.getInt(4 * a);
                    List<QueryResult> out = new ArrayList<>(eff);
                    List<String> fin = new ArrayList<>(eff);
                    for (int r = 0; r < eff; r++) {
                        String id = rowIds.get(a).get(oi.getInt(4 * (a * K + r)));
                        out.add(new QueryResult(id, od.getDouble(8 * (a * K + r))));
                        fin.add(id);
                    }
                    results.set(i, out);
                    returned[i] = eff;
                    if (i == nq - 1) finLast = fin;
                    if (attempt == 0 && (eff < K || decrypted[i] < 10 * K)) again.add(i);      // QSI:327-337,444-447
                }
                active = again;
                probes = 10;
            }
        } finally {                                                              // QSI:342-351
            index.clearProbeOverride();
            lastServerNs = System.nanoTime() - t0;
            lastDecryptNs = decryptNs;
            lastClientNs = 0;
            final int l = nq - 1;
            lastCandTotal = raw[l]; lastCandKept = kept[l]; lastCandDecrypted = decrypted[l]; lastReturned = returned[l];
            lastUniqueCandidates = unique[l]; lastFinalIds = finLast;
            touchedThisSession.clear();
            touchedThisSession.addAll(touched.get(l));
            if (reencTracker != null) for (Set<String> ts : touched) if (!ts.isEmpty()) reencTracker.record(new HashSet<>(ts));
        }
        return results;
    }

    // ---- accessors of QueryServiceImpl (QSI:83-87,417-474) ----------------------------------------------------------------
    public void setReencryptionTracker(ReencryptionTracker tr) { this.reencTracker = tr; }
    public List<String> getLastFinalResultIds() { return lastFinalIds; }
    public long getLastQueryDurationNs() { return lastServerNs; }
    public long getLastClientDurationNs() { return lastClientNs; }
    public long getLastDecryptNs() { return lastDecryptNs; }
    public int getLastCandTotal() { return lastCandTotal; }
    public int getLastCandKept() { return lastCandKept; }
    public Set<String> getLastCandidateIds() { return index.getLastTouchedIds(); }
    public int getLastCandDecrypted() { return lastCandDecrypted; }
    public int getLastReturned() { return lastReturned; }
    public QueryToken deriveToken(QueryToken base, int k) {
        if (tokenFactory == null) throw new IllegalStateException("QueryTokenFactory not available");
        return tokenFactory.derive(base, k);
    }
    public void setRefinementLimit(int limit) { refinementLimitOverride.set(limit); }
    public void clearRefinementLimit() { refinementLimitOverride.set(-1); }
    public int getEffectiveRefinementLimit(int defaultLimit) { int o = refinementLimitOverride.get(); return o > 0 ? o : defaultLimit; }
    public int getLastEffectiveRefinementLimit() { return lastEffectiveLimit; }
    public int getLastUniqueCandidates() { return lastUniqueCandidates; }
    public double getLastRefinementUtilization() { return lastEffectiveLimit > 0 ? (double) lastCandDecrypted / lastEffectiveLimit : 0.0; }
}
