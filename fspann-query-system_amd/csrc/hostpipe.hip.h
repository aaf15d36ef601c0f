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
#include "fspann_common.h"
#include "../host/pointstore.hpp"      // the point store + AES-GCM pool: pure host code, also built under TSan / ASan by the CPU suite

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
