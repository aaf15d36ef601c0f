// fspann_jni.cpp — JNI shim between com.fspann.gpu.FspannNative and the C ABI (include/fspann.h).
// Pure marshalling: every Java ByteBuffer must be DIRECT; null buffers become NULL pointers.
// Build (needs a JDK; none exists in the build container):
//   g++ -O2 -fPIC -shared -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -I../include \
//       -o libfspann_jni.so fspann_jni.cpp -L../fspann-query-system_amd -lfspann_hip
#include <jni.h>

#include <cstdint>

#include "fspann.h"

namespace {
inline void* addr(JNIEnv* env, jobject buf) { return buf ? env->GetDirectBufferAddress(buf) : nullptr; }
inline fspann_ctx* C(jlong h) { return reinterpret_cast<fspann_ctx*>(static_cast<intptr_t>(h)); }
}  // namespace

extern "C" {

JNIEXPORT jlong JNICALL Java_com_fspann_gpu_FspannNative_ctxCreate(JNIEnv* env, jclass, jint device, jintArray cfgArr) {
    jint v[11] = {0};
    jsize n = env->GetArrayLength(cfgArr);
    env->GetIntArrayRegion(cfgArr, 0, n < 11 ? n : 11, v);
    fspann_cfg cfg = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], 0};
    fspann_ctx* ctx = nullptr;
    int rc = fspann_ctx_create(device, &cfg, &ctx);
    if (rc != FSPANN_OK) {
        const char* cls = rc == FSPANN_E_ARG ? "java/lang/IllegalArgumentException"
                        : rc == FSPANN_E_NULL ? "java/lang/NullPointerException" : "java/lang/IllegalStateException";
        env->ThrowNew(env->FindClass(cls), fspann_last_error());
        return 0;
    }
    return static_cast<jlong>(reinterpret_cast<intptr_t>(ctx));
}

JNIEXPORT void JNICALL Java_com_fspann_gpu_FspannNative_ctxDestroy(JNIEnv*, jclass, jlong h) { fspann_ctx_destroy(C(h)); }

JNIEXPORT jstring JNICALL Java_com_fspann_gpu_FspannNative_lastError(JNIEnv* env, jclass) {
    return env->NewStringUTF(fspann_last_error());
}

JNIEXPORT jint JNICALL Java_com_fspann_gpu_FspannNative_setGFunctions(JNIEnv* env, jclass, jlong h, jobject a, jobject r, jobject w) {
    return fspann_set_gfunctions(C(h), static_cast<const double*>(addr(env, a)), static_cast<const double*>(addr(env, r)),
                                 static_cast<const double*>(addr(env, w)));
}

JNIEXPORT jint JNICALL Java_com_fspann_gpu_FspannNative_setIndex(JNIEnv* env, jclass, jlong h, jint td, jlong nParts, jobject mn,
                                                                 jobject mx, jobject rep, jobject off, jobject ids) {
    return fspann_set_index(C(h), td, nParts, static_cast<const int64_t*>(addr(env, mn)), static_cast<const int64_t*>(addr(env, mx)),
                            static_cast<const uint64_t*>(addr(env, rep)), static_cast<const int64_t*>(addr(env, off)),
                            static_cast<const int32_t*>(addr(env, ids)));
}

JNIEXPORT jint JNICALL Java_com_fspann_gpu_FspannNative_setIdMeta(JNIEnv* env, jclass, jlong h, jlong n, jobject jh, jobject del) {
    return fspann_set_id_meta(C(h), n, static_cast<const int32_t*>(addr(env, jh)), static_cast<const uint8_t*>(addr(env, del)));
}

JNIEXPORT jint JNICALL Java_com_fspann_gpu_FspannNative_finalizeIndex(JNIEnv*, jclass, jlong h) { return fspann_finalize(C(h)); }

JNIEXPORT jint JNICALL Java_com_fspann_gpu_FspannNative_encode(JNIEnv* env, jclass, jlong h, jlong nq, jobject q, jint dtype,
                                                               jobject codes, jobject hashes) {
    return fspann_encode(C(h), nq, addr(env, q), dtype, static_cast<uint64_t*>(addr(env, codes)),
                         static_cast<int32_t*>(addr(env, hashes)));
}

JNIEXPORT jint JNICALL Java_com_fspann_gpu_FspannNative_route(JNIEnv* env, jclass, jlong h, jlong nq, jobject codes, jint probes,
                                                              jint limit, jlong cap, jobject ids, jobject score, jobject count,
                                                              jobject kept, jobject raw) {
    return fspann_route(C(h), nq, static_cast<const uint64_t*>(addr(env, codes)), probes, limit, cap,
                        static_cast<int32_t*>(addr(env, ids)), static_cast<int32_t*>(addr(env, score)),
                        static_cast<int32_t*>(addr(env, count)), static_cast<int32_t*>(addr(env, kept)),
                        static_cast<int32_t*>(addr(env, raw)));
}

JNIEXPORT jlong JNICALL Java_com_fspann_gpu_FspannNative_routeMaxCandidates(JNIEnv*, jclass, jlong h, jint probes) {
    return fspann_route_max_candidates(C(h), probes);
}

JNIEXPORT jint JNICALL Java_com_fspann_gpu_FspannNative_refine(JNIEnv* env, jclass, jlong h, jlong nq, jobject q, jobject cand,
                                                               jint dtype, jlong B, jobject candIds, jobject candCount, jint k,
                                                               jobject outIds, jobject outDist, jobject outCount, jobject scored) {
    return fspann_refine(C(h), nq, addr(env, q), addr(env, cand), dtype, B, static_cast<const int32_t*>(addr(env, candIds)),
                         static_cast<const int32_t*>(addr(env, candCount)), k, static_cast<int32_t*>(addr(env, outIds)),
                         static_cast<double*>(addr(env, outDist)), static_cast<int32_t*>(addr(env, outCount)),
                         static_cast<int32_t*>(addr(env, scored)));
}

// Plaintext rows resident on the device (test / trusted-HBM deployments): set once, then refine by id.
JNIEXPORT jint JNICALL Java_com_fspann_gpu_FspannNative_storeSet(JNIEnv* env, jclass, jlong h, jlong n, jobject vectors, jint dtype) {
    return fspann_store_set(C(h), n, addr(env, vectors), dtype);
}

JNIEXPORT jint JNICALL Java_com_fspann_gpu_FspannNative_refineStore(JNIEnv* env, jclass, jlong h, jlong nq, jobject q, jint dtype, jlong B,
                                                                    jobject candIds, jobject candCount, jint k, jobject outIds,
                                                                    jobject outDist, jobject outCount, jobject scored) {
    return fspann_refine_store(C(h), nq, addr(env, q), dtype, B, static_cast<const int32_t*>(addr(env, candIds)),
                               static_cast<const int32_t*>(addr(env, candCount)), k, static_cast<int32_t*>(addr(env, outIds)),
                               static_cast<double*>(addr(env, outDist)), static_cast<int32_t*>(addr(env, outCount)),
                               static_cast<int32_t*>(addr(env, scored)));
}

JNIEXPORT jint JNICALL Java_com_fspann_gpu_FspannNative_setRouteMode(JNIEnv*, jclass, jlong h, jint mode) {
    return fspann_set_route_mode(C(h), mode);
}

}  // extern "C"
