// pointstore.hpp — the HOST half of Refine's stage B at batch scale: a packed point store in the reference's
// record format and an AES-256-GCM thread pool that opens the selected candidates (QSI:238-271 = loadPointIfActive +
// decryptFromPoint per candidate: PIS:717-724, crypto/AesGcmCryptoService.java:126-166,261-277 — 89-93 % of the reference's
// query latency), plus Rotate / Migrate / Retire as keymanagement/KeyRotationServiceImpl.java:215-334 runs them.
// Crypto and key derivation are restated bit for bit:
//   record   : key version, 12-byte IV, ciphertext = 8*dim bytes big-endian fp64 || 16-byte tag   (AesGcmCryptoService.java:55-112,240-259)
//   AAD      : "id:%s|v:%d|d:%d" with the decimal id                                              (common/EncryptedPoint.java:80-83)
//   K_v      : HMAC-SHA256(K_M, int32_be(v))                                                     (keymanagement/KeyManager.java:221-237)
//   migrate  : open with the record's version, seal with the current one and a fresh IV           (keymanagement/KeyRotationServiceImpl.java:215-289)
// PURE HOST C++17 (no HIP): it is compiled into libfspann_hip.so through csrc/hostpipe.hip.h AND, on its own, with
// g++ -fsanitize=thread / address,undefined by the CPU suite (tests/cpp/pointstore_stress.cpp) — the reader / writer protocol
// of a live Migrate (per-record version word: writer -1 -> rewrite -> new version; reader snapshot + re-check) is exactly the
// kind of code that wants a race detector, and GPU sanitizers do not exist on the pool.
// libcrypto (OpenSSL 3) is bound at run time like librccl: a deployment that brings its own decrypt loop (the JVM) never needs it.
#pragma once
#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>


namespace fspann {

struct CryptoApi {
    void* (*CTX_new)() = nullptr;
    void (*CTX_free)(void*) = nullptr;
    const void* (*aes_256_gcm)() = nullptr;
    int (*EncryptInit_ex)(void*, const void*, void*, const unsigned char*, const unsigned char*) = nullptr;
    int (*DecryptInit_ex)(void*, const void*, void*, const unsigned char*, const unsigned char*) = nullptr;
    int (*EncryptUpdate)(void*, unsigned char*, int*, const unsigned char*, int) = nullptr;
    int (*DecryptUpdate)(void*, unsigned char*, int*, const unsigned char*, int) = nullptr;
    int (*EncryptFinal_ex)(void*, unsigned char*, int*) = nullptr;
    int (*DecryptFinal_ex)(void*, unsigned char*, int*) = nullptr;
    int (*CTX_ctrl)(void*, int, int, void*) = nullptr;
    const void* (*sha256)() = nullptr;
    unsigned char* (*HMAC)(const void*, const void*, int, const unsigned char*, size_t, unsigned char*, unsigned int*) = nullptr;
    int (*RAND_bytes)(unsigned char*, int) = nullptr;
    void* handle = nullptr;
};

inline CryptoApi* crypto_api() {
    static CryptoApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("FSPANN_CRYPTO_LIB");
        const char* names[] = {env, "libcrypto.so.3", "libcrypto.so"};
        for (const char* nm : names) {
            if (!nm || !*nm) continue;
            void* h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (!h) continue;
            CryptoApi a;
            a.handle = h;
#define FSP_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name))
            FSP_SYM(CTX_new, "EVP_CIPHER_CTX_new"); FSP_SYM(CTX_free, "EVP_CIPHER_CTX_free"); FSP_SYM(aes_256_gcm, "EVP_aes_256_gcm");
            FSP_SYM(EncryptInit_ex, "EVP_EncryptInit_ex"); FSP_SYM(DecryptInit_ex, "EVP_DecryptInit_ex");
            FSP_SYM(EncryptUpdate, "EVP_EncryptUpdate"); FSP_SYM(DecryptUpdate, "EVP_DecryptUpdate");
            FSP_SYM(EncryptFinal_ex, "EVP_EncryptFinal_ex"); FSP_SYM(DecryptFinal_ex, "EVP_DecryptFinal_ex");
            FSP_SYM(CTX_ctrl, "EVP_CIPHER_CTX_ctrl"); FSP_SYM(sha256, "EVP_sha256"); FSP_SYM(HMAC, "HMAC"); FSP_SYM(RAND_bytes, "RAND_bytes");
#undef FSP_SYM
            if (a.CTX_new && a.CTX_free && a.aes_256_gcm && a.EncryptInit_ex && a.DecryptInit_ex && a.EncryptUpdate && a.DecryptUpdate &&
                a.EncryptFinal_ex && a.DecryptFinal_ex && a.CTX_ctrl && a.sha256 && a.HMAC && a.RAND_bytes) { api = a; return; }
            dlclose(h);
        }
    });
    return api.handle ? &api : nullptr;
}

constexpr int kGcmSetIvLen = 0x9, kGcmGetTag = 0x10, kGcmSetTag = 0x11;   // EVP_CTRL_AEAD_*
constexpr int kIvBytes = 12, kTagBytes = 16;
constexpr int kRecHeader = 8;        // bytes in front of a record's payload (iv || ciphertext || tag): the version word + 4 of padding, so the
                                     //   payload starts on an 8-byte boundary and is snapshotted in 8-byte words
inline size_t pointstore_stride(int dim) { return (kRecHeader + kIvBytes + 8 * static_cast<size_t>(dim) + kTagBytes + 7) & ~size_t(7); }

inline uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }

// The bytes of a record are read by query threads while a Migrate may be rewriting them; the per-record version word decides
// afterwards whether a snapshot is usable (seqlock).  The copies themselves therefore go through relaxed ATOMIC word accesses:
// a torn snapshot is expected and discarded, a data race in the C++ sense it is not (and ThreadSanitizer agrees).
// `shared` is at least 4-byte aligned (records start on 8-byte strides, payload at +kRecHeader), n a multiple of 4; 8-byte words
// where the address allows.
inline void copy_from_shared(void* dst, const void* shared, size_t n) {
    unsigned char* d = static_cast<unsigned char*>(dst);
    size_t i = 0;
    if ((reinterpret_cast<uintptr_t>(shared) & 7) == 0) {
        const uint64_t* s8 = static_cast<const uint64_t*>(shared);
        for (; i + 8 <= n; i += 8) { const uint64_t v = __atomic_load_n(s8 + i / 8, __ATOMIC_RELAXED); std::memcpy(d + i, &v, 8); }
    }
    const uint32_t* s = reinterpret_cast<const uint32_t*>(static_cast<const unsigned char*>(shared) + i);
    for (size_t k = 0; i + 4 <= n; i += 4, k++) { const uint32_t v = __atomic_load_n(s + k, __ATOMIC_RELAXED); std::memcpy(d + i, &v, 4); }
}
inline void copy_to_shared(void* shared, const void* src, size_t n) {
    const unsigned char* s = static_cast<const unsigned char*>(src);
    size_t i = 0;
    if ((reinterpret_cast<uintptr_t>(shared) & 7) == 0) {
        uint64_t* d8 = static_cast<uint64_t*>(shared);
        for (; i + 8 <= n; i += 8) { uint64_t v; std::memcpy(&v, s + i, 8); __atomic_store_n(d8 + i / 8, v, __ATOMIC_RELAXED); }
    }
    uint32_t* d = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(shared) + i);
    for (size_t k = 0; i + 4 <= n; i += 4, k++) { uint32_t v; std::memcpy(&v, s + i, 4); __atomic_store_n(d + k, v, __ATOMIC_RELAXED); }
}

// One pair of reusable cipher contexts per worker thread (the reference pays Cipher.getInstance per candidate).  The cipher is
// bound to a context ONCE and the key only when the version changes; a record then costs an IV reset + GCM over ~1 KB.
// (EVP_*Init_ex with the cipher argument re-resolves the implementation on every call in OpenSSL 3: measured 38 us per open.)
struct GcmWorker {
    CryptoApi* a;
    void* dctx;
    void* ectx;
    int dec_version = 0, enc_version = 0;     // key version currently loaded into each context (0: none)
    bool dec_ready = false, enc_ready = false;
    explicit GcmWorker(CryptoApi* api) : a(api), dctx(api->CTX_new()), ectx(api->CTX_new()) {
        dec_ready = dctx && a->DecryptInit_ex(dctx, a->aes_256_gcm(), nullptr, nullptr, nullptr) == 1 && a->CTX_ctrl(dctx, kGcmSetIvLen, kIvBytes, nullptr) == 1;
        enc_ready = ectx && a->EncryptInit_ex(ectx, a->aes_256_gcm(), nullptr, nullptr, nullptr) == 1 && a->CTX_ctrl(ectx, kGcmSetIvLen, kIvBytes, nullptr) == 1;
    }
    ~GcmWorker() { if (dctx) a->CTX_free(dctx); if (ectx) a->CTX_free(ectx); }
    GcmWorker(const GcmWorker&) = delete;
    GcmWorker& operator=(const GcmWorker&) = delete;
    bool set_dec_key(int version, const unsigned char* key) {
        if (!dec_ready || a->DecryptInit_ex(dctx, nullptr, nullptr, key, nullptr) != 1) { dec_version = 0; return false; }
        dec_version = version;
        return true;
    }
    bool set_enc_key(int version, const unsigned char* key) {
        if (!enc_ready || a->EncryptInit_ex(ectx, nullptr, nullptr, key, nullptr) != 1) { enc_version = 0; return false; }
        enc_version = version;
        return true;
    }
    // ct_tag = ciphertext || tag (javax.crypto doFinal layout).  false = tag mismatch (wrong key, AAD or corrupted record).
    bool open(const unsigned char* iv, const unsigned char* aad, int aad_len, const unsigned char* ct_tag, int ct_len, unsigned char* pt) {
        int n = 0;
        if (a->DecryptInit_ex(dctx, nullptr, nullptr, nullptr, iv) != 1) return false;
        if (aad_len > 0 && a->DecryptUpdate(dctx, nullptr, &n, aad, aad_len) != 1) return false;
        if (a->DecryptUpdate(dctx, pt, &n, ct_tag, ct_len) != 1) return false;
        if (a->CTX_ctrl(dctx, kGcmSetTag, kTagBytes, const_cast<unsigned char*>(ct_tag + ct_len)) != 1) return false;
        int m = 0;
        return a->DecryptFinal_ex(dctx, pt + n, &m) == 1;
    }
    bool seal(const unsigned char* iv, const unsigned char* aad, int aad_len, const unsigned char* pt, int pt_len, unsigned char* ct_tag) {
        int n = 0, m = 0;
        if (a->EncryptInit_ex(ectx, nullptr, nullptr, nullptr, iv) != 1) return false;
        if (aad_len > 0 && a->EncryptUpdate(ectx, nullptr, &n, aad, aad_len) != 1) return false;
        if (a->EncryptUpdate(ectx, ct_tag, &n, pt, pt_len) != 1) return false;
        if (a->EncryptFinal_ex(ectx, ct_tag + n, &m) != 1) return false;
        return a->CTX_ctrl(ectx, kGcmGetTag, kTagBytes, ct_tag + pt_len) == 1;
    }
};

}  // namespace fspann

// Packed point store: record h = { int32 key_version (0 = never written / deleted), uint8 iv[12], uint8 ct[8*dim + 16] }
// at a fixed stride, in memory.  The per-record version word is what makes a live Migrate safe next to readers: a writer
// flips it to -1, rewrites the record, then publishes the new version; a reader that sees -1 or a changed version retries.
namespace fspann {
// key material does not outlive its use: the compiler may not drop this store (OPENSSL_cleanse semantics)
inline void cleanse(void* p, size_t n) {
    volatile unsigned char* v = static_cast<volatile unsigned char*>(p);
    for (size_t i = 0; i < n; i++) v[i] = 0;
}
}  // namespace fspann

struct fspann_pointstore {
    ~fspann_pointstore() {
        fspann::cleanse(master, sizeof(master));
        for (auto& k : keys) if (!k.empty()) fspann::cleanse(k.data(), k.size());
    }
    int64_t n = 0;
    int dim = 0;
    size_t stride = 0;
    std::vector<unsigned char> mem;
    unsigned char master[32] = {0};
    bool have_master = false;
    std::atomic<int> current_version{1};
    std::mutex key_mu;
    std::vector<std::vector<unsigned char>> keys;   // K_v by version (derived on demand), index v
    std::vector<char> retired;                        // KeyManager retire: K_v no longer derivable
    std::atomic<long long> opened{0}, failed{0};

    unsigned char* rec(int64_t h) { return mem.data() + static_cast<size_t>(h) * stride; }
    std::atomic<int32_t>* ver(int64_t h) { return reinterpret_cast<std::atomic<int32_t>*>(rec(h)); }
    // K_v = HMAC-SHA256(K_M, int32_be(v)), first 32 bytes (KeyManager.java:221-237); false when retired / no master key
    bool key_for(int v, unsigned char out[32]) {
        if (v <= 0 || !have_master) return false;
        std::lock_guard<std::mutex> lk(key_mu);
        if (static_cast<size_t>(v) < retired.size() && retired[v]) return false;
        if (static_cast<size_t>(v) >= keys.size()) keys.resize(v + 1);
        if (keys[v].empty()) {
            fspann::CryptoApi* a = fspann::crypto_api();
            if (!a) return false;
            const unsigned char salt[4] = {static_cast<unsigned char>(v >> 24), static_cast<unsigned char>(v >> 16), static_cast<unsigned char>(v >> 8),
                                           static_cast<unsigned char>(v)};
            unsigned char md[64];
            unsigned int mdlen = 0;
            if (!a->HMAC(a->sha256(), master, 32, salt, 4, md, &mdlen) || mdlen < 32) return false;
            keys[v].assign(md, md + 32);
        }
        std::memcpy(out, keys[v].data(), 32);
        return true;
    }
};

namespace fspann {

// Writers exclude each other on the record's version word: take it from whatever it is to -1 ("being written"); a record another
// writer holds is waited for.  (An unconditional store of -1 let an encryptToPoint of an existing id and a Migrate of the same id
// write the same bytes at once: a record neither key could open — found by tests/cpp/pointstore_stress.cpp.)
inline int32_t acquire_record(fspann_pointstore* ps, int64_t h) {
    for (long spin = 0;; spin++) {
        int32_t cur = ps->ver(h)->load(std::memory_order_acquire);
        if (cur != -1 && ps->ver(h)->compare_exchange_weak(cur, -1, std::memory_order_acq_rel)) return cur;
        if (spin < 16) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// "id:%s|v:%d|d:%d" (EncryptedPoint.java:80-83, id = Long.toString(handle)) without snprintf: it is built once per record on the
// open path (tools/micro/open_bench.cpp: 0.10 us of a 0.38 us cipher call through snprintf)
inline int put_dec(char* p, long long v) {
    char tmp[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ull - static_cast<unsigned long long>(v) : static_cast<unsigned long long>(v);
    do { tmp[n++] = static_cast<char>('0' + u % 10); u /= 10; } while (u);
    int k = 0;
    if (v < 0) p[k++] = '-';
    while (n) p[k++] = tmp[--n];
    return k;
}
inline int aad_for(char* buf, size_t cap, int64_t handle, int version, int dim) {
    if (cap < 80) return snprintf(buf, cap, "id:%lld|v:%d|d:%d", static_cast<long long>(handle), version, dim);
    int k = 0;
    buf[k++] = 'i'; buf[k++] = 'd'; buf[k++] = ':';
    k += put_dec(buf + k, handle);
    buf[k++] = '|'; buf[k++] = 'v'; buf[k++] = ':';
    k += put_dec(buf + k, version);
    buf[k++] = '|'; buf[k++] = 'd'; buf[k++] = ':';
    k += put_dec(buf + k, dim);
    buf[k] = 0;
    return k;
}

// Run fn(worker_index, begin, end) over [0, n) on `threads` threads (contiguous ranges handed out in blocks of `grain`).
template <class F>
inline void parallel_blocks(int64_t n, int threads, int64_t grain, F&& fn) {
    threads = std::max(1, threads);
    if (threads == 1 || n <= grain) { fn(0, int64_t(0), n); return; }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> pool;
    auto body = [&](int w) {
        for (;;) {
            const int64_t b = next.fetch_add(grain);
            if (b >= n) return;
            fn(w, b, std::min(n, b + grain));
        }
    };
    for (int w = 1; w < threads; w++) pool.emplace_back(body, w);
    body(0);
    for (auto& t : pool) t.join();
}

// Seal handles [h0, h0 + cnt) with the current key version.  src row i = vector of handle h0 + i.
template <typename T>
inline int pointstore_encrypt(fspann_pointstore* ps, int64_t h0, int64_t cnt, const T* src, int threads, std::atomic<long long>* bad) {
    CryptoApi* a = crypto_api();
    const int v = ps->current_version.load();
    unsigned char key[32];
    if (!ps->key_for(v, key)) return -1;
    const int dim = ps->dim, ptlen = 8 * dim;
    // every IV comes from ONE RAND_bytes call on the calling thread: OpenSSL instantiates a DRBG per thread on first use, and
    // with short-lived worker threads that instantiation (serialised on the parent DRBG) was 10 s for 1 M records
    std::vector<unsigned char> ivs(static_cast<size_t>(cnt) * kIvBytes);
    for (size_t off = 0; off < ivs.size(); off += (1u << 30)) {
        const int len = static_cast<int>(std::min<size_t>(ivs.size() - off, 1u << 30));
        if (a->RAND_bytes(ivs.data() + off, len) != 1) return -2;
    }
    parallel_blocks(cnt, threads, 1024, [&](int, int64_t b, int64_t e) {
        GcmWorker w(a);
        if (!w.set_enc_key(v, key)) { (*bad) += e - b; return; }
        std::vector<unsigned char> pt(ptlen), sealed(kIvBytes + ptlen + kTagBytes);
        char aad[96];
        for (int64_t i = b; i < e; i++) {
            const int64_t h = h0 + i;
            for (int j = 0; j < dim; j++) {       // serializeVector: big-endian IEEE-754 doubles (the float -> double widening is exact)
                const double x = static_cast<double>(src[i * dim + j]);
                uint64_t bits;
                std::memcpy(&bits, &x, 8);
                bits = bswap64(bits);
                std::memcpy(pt.data() + 8 * j, &bits, 8);
            }
            unsigned char* r = ps->rec(h);
            const unsigned char* iv = ivs.data() + static_cast<size_t>(i) * kIvBytes;
            const int al = aad_for(aad, sizeof(aad), h, v, dim);
            std::memcpy(sealed.data(), iv, kIvBytes);
            if (!w.seal(iv, reinterpret_cast<const unsigned char*>(aad), al, pt.data(), ptlen, sealed.data() + kIvBytes)) { (*bad)++; continue; }
            (void)acquire_record(ps, h);                               // being written: -1 is visible before the record's bytes change
            copy_to_shared(r + kRecHeader, sealed.data(), sealed.size());
            ps->ver(h)->store(v, std::memory_order_release);
        }
    });
    return 0;
}

// deserializeVector (AesGcmCryptoService.java:261-277): dim big-endian IEEE-754 doubles -> host doubles / the fp32 staging type
inline void decode_row(const unsigned char* pt, int dim, double* out) {
    for (int j = 0; j < dim; j++) {
        uint64_t bits;
        std::memcpy(&bits, pt + 8 * j, 8);
        bits = bswap64(bits);
        std::memcpy(out + j, &bits, 8);
    }
}
inline void decode_row(const unsigned char* pt, int dim, float* out) {       // (double) -> float: the narrowing QSI never does — exact for
    for (int j = 0; j < dim; j++) {                                           //   fvecs-derived data, the fp32-staged block of SURVEY §8d)
        uint64_t bits;
        std::memcpy(&bits, pt + 8 * j, 8);
        bits = bswap64(bits);
        double x;
        std::memcpy(&x, &bits, 8);
        out[j] = static_cast<float>(x);
    }
}
#if defined(__x86_64__)
// the same with 32-byte vectors: byte-reverse four doubles with one shuffle, narrow with one convert (chosen at run time)
__attribute__((target("avx2"))) inline void decode_row_avx2(const unsigned char* pt, int dim, float* out) {
    typedef long long v4di __attribute__((vector_size(32)));
    typedef char v32qi __attribute__((vector_size(32)));
    typedef double v4df __attribute__((vector_size(32)));
    typedef float v4sf __attribute__((vector_size(16)));
    const v32qi rev = {7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8};
    int j = 0;
    for (; j + 4 <= dim; j += 4) {
        v4di raw;
        std::memcpy(&raw, pt + 8 * j, 32);
        const v32qi sw = __builtin_ia32_pshufb256(reinterpret_cast<v32qi>(raw), rev);
        const v4sf f = __builtin_ia32_cvtpd2ps256(reinterpret_cast<v4df>(sw));
        std::memcpy(out + j, &f, 16);
    }
    if (j < dim) decode_row(pt + 8 * j, dim - j, out + j);
}
__attribute__((target("avx2"))) inline void decode_row_avx2(const unsigned char* pt, int dim, double* out) {
    typedef long long v4di __attribute__((vector_size(32)));
    typedef char v32qi __attribute__((vector_size(32)));
    const v32qi rev = {7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8};
    int j = 0;
    for (; j + 4 <= dim; j += 4) {
        v4di raw;
        std::memcpy(&raw, pt + 8 * j, 32);
        const v32qi sw = __builtin_ia32_pshufb256(reinterpret_cast<v32qi>(raw), rev);
        std::memcpy(out + j, &sw, 32);
    }
    if (j < dim) decode_row(pt + 8 * j, dim - j, out + j);
}
inline bool cpu_has_avx2() { static const bool v = __builtin_cpu_supports("avx2"); return v; }
#else
inline bool cpu_has_avx2() { return false; }
template <typename T> inline void decode_row_avx2(const unsigned char* pt, int dim, T* out) { decode_row(pt, dim, out); }
#endif
template <typename T> inline void decode_row_fast(const unsigned char* pt, int dim, T* out) {
    if (cpu_has_avx2()) decode_row_avx2(pt, dim, out); else decode_row(pt, dim, out);
}

// decryptFromPoint of one record into `out` (dim doubles, host byte order).  Retries while a writer holds the record.
template <typename TOut>
inline bool pointstore_open_one(fspann_pointstore* ps, GcmWorker& w, int64_t h, std::vector<unsigned char>& scratch, TOut* out, int* version_out) {
    const int dim = ps->dim, ctlen = 8 * dim;
    if (h < 0 || h >= ps->n) return false;
    // A writer holds a record (version word -1) only while it copies ~1 KB, but on an oversubscribed host it can be
    // descheduled in the middle: the wait is bounded by TIME (2 s), not by a spin count — a reader that gave up after 1000
    // yields reported a live record as failed (seen once in the 1 M-record rotate + migrate test on a 4-core share).
    const auto t_start = std::chrono::steady_clock::now();
    int tag_failures = 0;
    for (long attempt = 0;; attempt++) {
        if (attempt > 64 && (attempt & 63) == 0 &&
            std::chrono::steady_clock::now() - t_start > std::chrono::seconds(2)) return false;
        const int v = ps->ver(h)->load(std::memory_order_acquire);
        if (v == 0) return false;                      // never written / deleted: loadPointIfActive() == null
        if (v < 0) {
            if (attempt < 16) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
            continue;
        }
        if (w.dec_version != v) {     // derivation + the store's key mutex only when the version changes (one batch's worth of life)
            unsigned char key[32];
            if (!ps->key_for(v, key) || !w.set_dec_key(v, key)) return false;        // retired key: the old ciphertext is unreadable by design
        }
        copy_from_shared(scratch.data(), ps->rec(h) + kRecHeader, kIvBytes + ctlen + kTagBytes);   // snapshot, then re-check the version
        std::atomic_thread_fence(std::memory_order_acquire);                            // the copy's reads stay ahead of the re-check
        if (ps->ver(h)->load(std::memory_order_relaxed) != v) continue;
        char aad[96];
        const int al = aad_for(aad, sizeof(aad), h, v, dim);
        unsigned char* pt = scratch.data() + kIvBytes + ctlen + kTagBytes;
        if (!w.open(scratch.data(), reinterpret_cast<const unsigned char*>(aad), al, scratch.data() + kIvBytes, ctlen, pt)) {
            if (ps->ver(h)->load(std::memory_order_acquire) != v) continue;             // rewritten since: the snapshot is stale, not corrupt
            // The version word cannot tell a rewrite that publishes the SAME version (encryptToPoint of an existing id under the
            // current key: v -> -1 -> v) from no rewrite at all: a torn snapshot then fails the tag with v unchanged.  Take a fresh
            // snapshot a few times before calling the record unreadable (a really corrupt record fails every time).
            if (++tag_failures <= 3) continue;
            return false;
        }
        decode_row_fast(pt, dim, out);                 // deserializeVector
        if (version_out) *version_out = v;
        return true;
    }
}

// KeyRotationServiceImpl.reencryptTouched (:215-289): records older than the current version are opened with THEIR key and
// sealed again with the current one under a fresh IV (and the new version in the AAD); failures are skipped silently (:274-276).
// Returns 0, -1 (current key not derivable) or -2 (RAND_bytes failed); *done = records moved.
inline int pointstore_reencrypt(fspann_pointstore* ps, const int32_t* handles, int64_t cnt, int threads, long long* done_out) {
    CryptoApi* a = crypto_api();
    const int target = ps->current_version.load();
    unsigned char tkey[32];
    if (!ps->key_for(target, tkey)) return -1;
    std::atomic<long long> done{0};
    const int dim = ps->dim, ptlen = 8 * dim;
    std::vector<unsigned char> ivs(static_cast<size_t>(cnt) * kIvBytes);     // fresh IVs, one RAND_bytes call on this thread
    if (cnt > 0 && a->RAND_bytes(ivs.data(), static_cast<int>(std::min<size_t>(ivs.size(), 1u << 30))) != 1) { cleanse(tkey, 32); return -2; }
    parallel_blocks(cnt, threads, 256, [&](int, int64_t b, int64_t e) {
        GcmWorker w(a);
        if (!w.set_enc_key(target, tkey)) return;
        std::vector<unsigned char> scratch(kIvBytes + 16 * static_cast<size_t>(dim) + kTagBytes + 64), pt(ptlen), sealed(ptlen + kTagBytes);
        std::vector<double> row(dim);
        char aad[96];
        for (int64_t i = b; i < e; i++) {
            const int64_t h = handles[i];
            if (h < 0 || h >= ps->n) continue;
            int oldv = 0;
            if (!pointstore_open_one(ps, w, h, scratch, row.data(), &oldv)) continue;     // forward-secure skip
            if (oldv >= target) continue;                                                 // already upgraded
            for (int j = 0; j < dim; j++) { uint64_t bits; std::memcpy(&bits, &row[j], 8); bits = bswap64(bits); std::memcpy(pt.data() + 8 * j, &bits, 8); }
            const unsigned char* iv = ivs.data() + static_cast<size_t>(i) * kIvBytes;
            const int al = aad_for(aad, sizeof(aad), h, target, dim);
            if (!w.seal(iv, reinterpret_cast<const unsigned char*>(aad), al, pt.data(), ptlen, sealed.data())) continue;
            int32_t expect = oldv;
            if (!ps->ver(h)->compare_exchange_strong(expect, -1, std::memory_order_acq_rel)) continue;   // someone else rewrote it meanwhile
            unsigned char* r = ps->rec(h);
            copy_to_shared(r + kRecHeader, iv, kIvBytes);
            copy_to_shared(r + kRecHeader + kIvBytes, sealed.data(), static_cast<size_t>(ptlen) + kTagBytes);
            ps->ver(h)->store(target, std::memory_order_release);
            done++;
        }
    });
    cleanse(tkey, 32);
    if (done_out) *done_out = done.load();
    return 0;
}

// QSI stage B, host part, for a batch: every (query, j < count[q]) candidate is loaded + opened; rows that fail are
// skipped (QSI:240-270 swallows per-candidate failures) and the survivors are PACKED to the front of the query's block
// in F_q order.  dst = [nq][B][dim] of TOut (float: the fp32-staged block of SURVEY §8d; exact for fvecs-derived data).
template <typename TOut>
inline void pointstore_open_batch(fspann_pointstore* ps, int64_t nq, int64_t B, const int32_t* ids, const int32_t* count, TOut* dst, int32_t* out_ids,
                                  int32_t* out_count, int threads) {
    CryptoApi* a = crypto_api();
    const int dim = ps->dim;
    // per-THREAD state (cipher contexts, scratch), created on the thread's first block and kept for the whole call
    struct PerThread { GcmWorker w; std::vector<unsigned char> scratch; std::vector<double> row;
                       PerThread(CryptoApi* a, int dim) : w(a), scratch(kIvBytes + 16 * static_cast<size_t>(dim) + kTagBytes + 64), row(dim) {} };
    std::vector<std::unique_ptr<PerThread>> state(static_cast<size_t>(std::max(1, threads)));
    parallel_blocks(nq, threads, 2, [&](int wi, int64_t qb, int64_t qe) {
        if (!state[wi]) state[wi].reset(new PerThread(a, dim));
        GcmWorker& w = state[wi]->w;
        std::vector<unsigned char>& scratch = state[wi]->scratch;
        long long okc = 0, badc = 0;
        for (int64_t q = qb; q < qe; q++) {
            const int c = std::max(0, std::min<int>(count[q], static_cast<int>(B)));
            int kept = 0;
            for (int j = 0; j < c; j++) {
                const int32_t id = ids[q * B + j];
                if (j + 1 < c) {                                  // F_q's records are scattered over the store: the next one is requested now
                    const int32_t nx = ids[q * B + j + 1];
                    if (nx >= 0 && nx < ps->n) { const unsigned char* r = ps->rec(nx); for (size_t off = 0; off < ps->stride; off += 64) __builtin_prefetch(r + off, 0, 0); }
                }
                // decoded straight into the query's block (a failed open leaves the slot to the next survivor)
                if (!pointstore_open_one(ps, w, id, scratch, dst + (q * B + kept) * dim, nullptr)) { badc++; continue; }
                out_ids[q * B + kept] = id;
                kept++;
                okc++;
            }
            for (int j = kept; j < B; j++) out_ids[q * B + j] = -1;
            out_count[q] = kept;
        }
        ps->opened += okc;
        ps->failed += badc;
    });
}

}  // namespace fspann

