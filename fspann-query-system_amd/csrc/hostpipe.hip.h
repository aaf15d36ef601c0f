// hostpipe.hip.h — the HOST side of Refine's stage B at batch scale: a packed point store, an AES-256-GCM
// thread pool that opens the selected candidates straight into pinned staging buffers, and a three-stage pipeline
// (Route of batch i+1 | decrypt of batch i | H2D + Refine of batch i-1) over one fspann context.
//
// What it replaces in the reference, per candidate (QSI:238-271): loadPointIfActive = one RocksDB get + one
// Java-deserialised .point file per id (PIS:717-724, common/RocksDBMetadataManager.java:530-544), then decryptFromPoint =
// Cipher.getInstance + AES-GCM open + big-endian fp64 decode (crypto/AesGcmCryptoService.java:126-166,261-277) — 89-93 % of
// the reference's query latency (/root/reference/README.md:309-311).  Crypto and key derivation are restated bit for bit:
//   record   : key version, 12-byte IV, ciphertext = 8*dim bytes big-endian fp64 || 16-byte tag   (AesGcmCryptoService.java:55-112,240-259)
//   AAD      : "id:%s|v:%d|d:%d" with the decimal id                                              (common/EncryptedPoint.java:80-83)
//   K_v      : HMAC-SHA256(K_M, int32_be(v))                                                     (keymanagement/KeyManager.java:221-237)
//   migrate  : open with the record's version, seal with the current one and a fresh IV           (keymanagement/KeyRotationServiceImpl.java:215-289)
// north_star keeps decrypt on the host: nothing here runs on the GPU except what fspann_refine_dev already did.
//
// libcrypto (OpenSSL 3) is bound at run time like librccl: a deployment that brings its own decrypt loop (the JVM) never
// needs it.
#pragma once
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "fspann_common.h"

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

inline uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }

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
struct fspann_pointstore {
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

inline int aad_for(char* buf, size_t cap, int64_t handle, int version, int dim) {   // EncryptedPoint.java:80-83, id = Long.toString(handle)
    return snprintf(buf, cap, "id:%lld|v:%d|d:%d", static_cast<long long>(handle), version, dim);
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
        std::vector<unsigned char> pt(ptlen);
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
            ps->ver(h)->store(-1, std::memory_order_relaxed);         // being written ...
            std::atomic_thread_fence(std::memory_order_release);      // ... and the record's bytes change only after that is visible
            const unsigned char* iv = ivs.data() + static_cast<size_t>(i) * kIvBytes;
            const int al = aad_for(aad, sizeof(aad), h, v, dim);
            std::memcpy(r + 4, iv, kIvBytes);
            if (!w.seal(iv, reinterpret_cast<const unsigned char*>(aad), al, pt.data(), ptlen, r + 4 + kIvBytes)) { (*bad)++; continue; }
            ps->ver(h)->store(v, std::memory_order_release);
        }
    });
    return 0;
}

// decryptFromPoint of one record into `out` (dim doubles, host byte order).  Retries while a writer holds the record.
inline bool pointstore_open_one(fspann_pointstore* ps, GcmWorker& w, int64_t h, std::vector<unsigned char>& scratch, double* out, int* version_out) {
    const int dim = ps->dim, ctlen = 8 * dim;
    if (h < 0 || h >= ps->n) return false;
    // A writer holds a record (version word -1) only while it copies ~1 KB, but on an oversubscribed host it can be
    // descheduled in the middle: the wait is bounded by TIME (2 s), not by a spin count — a reader that gave up after 1000
    // yields reported a live record as failed (seen once in the 1 M-record rotate + migrate test on a 4-core share).
    const auto t_start = std::chrono::steady_clock::now();
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
        std::memcpy(scratch.data(), ps->rec(h) + 4, kIvBytes + ctlen + kTagBytes);     // snapshot, then re-check the version
        std::atomic_thread_fence(std::memory_order_acquire);                            // the copy's reads stay ahead of the re-check
        if (ps->ver(h)->load(std::memory_order_relaxed) != v) continue;
        char aad[96];
        const int al = aad_for(aad, sizeof(aad), h, v, dim);
        unsigned char* pt = scratch.data() + kIvBytes + ctlen + kTagBytes;
        if (!w.open(scratch.data(), reinterpret_cast<const unsigned char*>(aad), al, scratch.data() + kIvBytes, ctlen, pt)) {
            if (ps->ver(h)->load(std::memory_order_acquire) != v) continue;             // rewritten since: the snapshot is stale, not corrupt
            return false;
        }
        for (int j = 0; j < dim; j++) {                // deserializeVector
            uint64_t bits;
            std::memcpy(&bits, pt + 8 * j, 8);
            bits = bswap64(bits);
            std::memcpy(out + j, &bits, 8);
        }
        if (version_out) *version_out = v;
        return true;
    }
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
        std::vector<double>& row = state[wi]->row;
        long long okc = 0, badc = 0;
        for (int64_t q = qb; q < qe; q++) {
            const int c = std::max(0, std::min<int>(count[q], static_cast<int>(B)));
            int kept = 0;
            for (int j = 0; j < c; j++) {
                const int32_t id = ids[q * B + j];
                if (!pointstore_open_one(ps, w, id, scratch, row.data(), nullptr)) { badc++; continue; }
                TOut* o = dst + (q * B + kept) * dim;
                for (int t = 0; t < dim; t++) o[t] = static_cast<TOut>(row[t]);
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

// ---- three-stage pipeline over one context ----------------------------------------------------------------------------
//   stage A (GPU)  : queries H2D, encode, Route (limit = B), F_q ids + counts D2H
//   stage B (host) : pointstore_open_batch into the batch's PINNED candidate block
//   stage C (GPU)  : candidate block H2D, fspann_refine_dev, top-k D2H
// Three batches are in flight (one per stage); each stage is a thread, batches move through bounded queues; the GPU work
// of A and C shares the context's stream (both are short next to B: 262 144 AES-GCM opens per batch).
struct fspann_pipeline {
    struct Slot {
        int64_t nq = 0;
        uint64_t ticket = 0;
        float* q_pin = nullptr;          // [nq_max][dim]
        int32_t* sel_pin = nullptr;      // [nq_max][B]
        int32_t* cnt_pin = nullptr;      // [nq_max]
        float* cand_pin = nullptr;       // [nq_max][B][dim]
        int32_t* ids_pin = nullptr;      // [nq_max][B] packed ids after decrypt
        int32_t* kcnt_pin = nullptr;     // [nq_max]
        int32_t* out_ids_pin = nullptr;  // [nq_max][k]
        double* out_dist_pin = nullptr;
        int32_t* out_cnt_pin = nullptr;
        void *q_dev = nullptr, *codes_dev = nullptr, *sel_dev = nullptr, *cnt_dev = nullptr, *cand_dev = nullptr, *ids_dev = nullptr, *kcnt_dev = nullptr,
             *oi_dev = nullptr, *od_dev = nullptr, *oc_dev = nullptr, *bad_dev = nullptr;
        int rc = 0;
        int64_t unmodelled = 0;          // queries of this batch the host model could not finish either (count stays -1: empty result)
        double t_route_ms = 0, t_decrypt_ms = 0, t_refine_ms = 0;
    };
    fspann_ctx* ctx = nullptr;
    fspann_pointstore* ps = nullptr;
    int64_t nq_max = 0, B = 0;
    int k = 0, threads = 1;
    static constexpr int kSlots = 4;
    Slot slot[kSlots];
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> free_q, qa, qb, qc, done_q;     // slot indices waiting for: a producer, stage A, B, C, the consumer
    bool stop = false;
    uint64_t next_ticket = 1;
    std::thread ta, tb, tc;
    std::mutex gpu_mu;                               // stage A and C take turns on the context
    double sum_route_ms = 0, sum_decrypt_ms = 0, sum_refine_ms = 0;
    long long batches = 0;
};
