package com.fspann.gpu;

import java.nio.ByteBuffer;

/**
 * JNI binding of libfspann_hip.so (include/fspann.h) — pure marshalling, one native method per C entry point the JVM adapter uses
 * (the device-pointer *_dev variants, index save/load and the native Setup are reached from C++/Python hosts).
 *
 * All buffers are DIRECT ByteBuffers in native byte order (the C side reads them in place; no copies on the Java
 * side).  Every method returns the C return code; {@link #check(int)} turns it into the exception class the
 * reference itself would throw (FSPANN_E_STATE -> IllegalStateException, FSPANN_E_ARG -> IllegalArgumentException,
 * FSPANN_E_NULL -> NullPointerException).
 *
 * NOT compiled in the build container (no JDK there); compile-gated on JAVA_HOME by jni/Makefile.
 */
public final class FspannNative {
    static { System.loadLibrary("fspann_jni"); }   // links libfspann_hip.so

    private FspannNative() {}

    public static final int F32 = 0, F64 = 1;

    /** cfg = {tables, divisions, m, lambda, dim, blockSize, defaultProbes, probeOverride, maxGlobalCandidates,
     *  refinementLimit, hammingPrefilterThreshold}; returns the context handle or throws. */
    public static native long ctxCreate(int device, int[] cfg);
    public static native void ctxDestroy(long ctx);
    public static native String lastError();

    /** alpha[T*D][m][dim], r[T*D][m], omega[T*D][m] as fp64 (Coding.GFunction fields, idx/Coding.java:52-97). */
    public static native int setGFunctions(long ctx, ByteBuffer alpha, ByteBuffer r, ByteBuffer omega);
    /** One (t,d) table of GreedyPartitioner.Partition as SoA (idx/GreedyPartitioner.java:13-32). */
    public static native int setIndex(long ctx, int td, long nParts, ByteBuffer minKey, ByteBuffer maxKey,
                                      ByteBuffer rep, ByteBuffer idOff, ByteBuffer ids);
    /** javaHash[h] = id.hashCode(); deleted[h] != 0 mirrors metadata.isDeleted(id); either may be null. */
    public static native int setIdMeta(long ctx, long nIds, ByteBuffer javaHash, ByteBuffer deleted);
    public static native int finalizeIndex(long ctx);

    /** Coding.C for all T*D GFunctions: q = [nq][dim] fp64, codes = [nq][T*D][W] u64 (BitSet.toLongArray layout). */
    public static native int encode(long ctx, long nq, ByteBuffer q, int dtype, ByteBuffer codes, ByteBuffer hashes);
    /** PIS.lookupCandidatesWithScores / lookupCandidateIds / QSI stage A.5, by `limit`. */
    public static native int route(long ctx, long nq, ByteBuffer codes, int probeOverride, int limit, long cap,
                                   ByteBuffer ids, ByteBuffer score, ByteBuffer count, ByteBuffer kept, ByteBuffer rawSeen);
    public static native long routeMaxCandidates(long ctx, int probeOverride);
    /** QSI stage B (distances) + C on packed decrypted candidates cand = [nq][B][dim]. */
    public static native int refine(long ctx, long nq, ByteBuffer q, ByteBuffer cand, int dtype, long B,
                                    ByteBuffer candIds, ByteBuffer candCount, int k,
                                    ByteBuffer outIds, ByteBuffer outDist, ByteBuffer outCount, ByteBuffer scored);

    /** Plaintext rows resident on the device (tests / trusted-HBM deployments), then stage B/C by id without a staging copy. */
    public static native int storeSet(long ctx, long n, ByteBuffer vectors, int dtype);
    public static native int refineStore(long ctx, long nq, ByteBuffer q, int dtype, long B, ByteBuffer candIds, ByteBuffer candCount,
                                         int k, ByteBuffer outIds, ByteBuffer outDist, ByteBuffer outCount, ByteBuffer scored);
    /** 0 auto, 1 full select, 2 bounded select whenever legal; route(..., kept = null, rawSeen = null) enables the bounded one. */
    public static native int setRouteMode(long ctx, int mode);

    public static void check(int rc) {
        if (rc == 0) return;
        String msg = lastError();
        switch (rc) {
            case -1: throw new IllegalStateException(msg);
            case -2: throw new IllegalArgumentException(msg);
            case -3: throw new NullPointerException(msg);
            case -5: throw new OutOfMemoryError(msg);
            default: throw new RuntimeException("fspann(" + rc + "): " + msg);
        }
    }
}
