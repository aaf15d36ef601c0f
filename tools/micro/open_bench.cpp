// Dev micro-benchmark (pure host, no GPU): what ONE thread pays to open one record of the packed point store
// (pointstore_open_one = QSI stage B's host half per candidate, cry/AesGcmCryptoService.java:126-166,261-277) against the bare
// cipher — EVP AES-256-GCM over the same 1024-byte ciphertext + tag + AAD with nothing around it — and where the rest goes.
// build: g++ -O2 -std=c++17 -pthread tools/micro/open_bench.cpp -ldl -o tools/micro/open_bench ; run: tools/micro/open_bench [records] [threads]
#include <chrono>
#include <cstdio>
#include <random>
#include "../../fspann-query-system_amd/host/pointstore.hpp"

using namespace fspann;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 262144;
    const int threads = argc > 2 ? atoi(argv[2]) : 1;
    const int dim = 128;
    CryptoApi* a = crypto_api();
    if (!a) { printf("no libcrypto\n"); return 1; }
    fspann_pointstore ps;
    ps.n = n; ps.dim = dim;
    ps.stride = pointstore_stride(dim);
    ps.mem.assign(static_cast<size_t>(n) * ps.stride, 0);
    for (int i = 0; i < 32; i++) ps.master[i] = static_cast<unsigned char>(i * 7 + 1);
    ps.have_master = true;
    std::vector<float> X(static_cast<size_t>(n) * dim);
    std::mt19937 rng(1);
    for (auto& x : X) x = static_cast<float>(static_cast<int>(rng() % 256));
    std::atomic<long long> bad{0};
    double t0 = now();
    pointstore_encrypt<float>(&ps, 0, n, X.data(), 8, &bad);
    printf("encrypt %lld records: %.2f s (bad %lld)\n", (long long)n, now() - t0, bad.load());
    std::vector<int32_t> order(n);
    for (int64_t i = 0; i < n; i++) order[i] = static_cast<int32_t>(i);
    std::shuffle(order.begin(), order.end(), rng);      // F_q ids are scattered over the store

    // ---- bare cipher: one context, key set once, per message: IV, AAD, update, tag, final -------------------------------------------
    {
        GcmWorker w(a);
        unsigned char key[32];
        ps.key_for(1, key);
        w.set_dec_key(1, key);
        std::vector<unsigned char> pt(8 * dim + 64);
        const unsigned char* r = ps.rec(order[0]) + kRecHeader;
        char aad[96];
        const int al = aad_for(aad, sizeof(aad), order[0], 1, dim);
        const int reps = 200000;
        t0 = now();
        int ok = 0;
        for (int i = 0; i < reps; i++) ok += w.open(r, reinterpret_cast<const unsigned char*>(aad), al, r + kIvBytes, 8 * dim, pt.data());
        const double dt = now() - t0;
        printf("bare EVP AES-256-GCM open, same 1024 B + tag + %d B AAD, hot: %.3f us per message (%.2f GB/s), ok %d/%d\n", al, dt / reps * 1e6,
               (8.0 * dim + 16) * reps / dt / 1e9, ok, reps);
        // components
        int nn = 0;
        t0 = now();
        for (int i = 0; i < reps; i++) a->DecryptInit_ex(w.dctx, nullptr, nullptr, nullptr, r);
        printf("  EVP_DecryptInit_ex(iv)      %.3f us\n", (now() - t0) / reps * 1e6);
        t0 = now();
        for (int i = 0; i < reps; i++) { a->DecryptInit_ex(w.dctx, nullptr, nullptr, nullptr, r); a->DecryptUpdate(w.dctx, nullptr, &nn, reinterpret_cast<const unsigned char*>(aad), al); }
        printf("  + AAD update                %.3f us\n", (now() - t0) / reps * 1e6);
        t0 = now();
        for (int i = 0; i < reps; i++) { a->DecryptInit_ex(w.dctx, nullptr, nullptr, nullptr, r); a->DecryptUpdate(w.dctx, nullptr, &nn, reinterpret_cast<const unsigned char*>(aad), al);
                                         a->DecryptUpdate(w.dctx, pt.data(), &nn, r + kIvBytes, 8 * dim); }
        printf("  + update(1024 B)            %.3f us\n", (now() - t0) / reps * 1e6);
        char buf[96];
        t0 = now();
        long s = 0;
        for (int i = 0; i < reps; i++) s += aad_for(buf, sizeof(buf), order[i % n], 1, dim);
        printf("  AAD string                  %.3f us (%ld)\n", (now() - t0) / reps * 1e6, s);
        std::vector<double> row(dim);
        t0 = now();
        for (int i = 0; i < reps; i++) decode_row(pt.data(), dim, row.data());
        printf("  big-endian fp64 decode      %.3f us\n", (now() - t0) / reps * 1e6);
    }
    // ---- the store's open, one thread, scattered records ---------------------------------------------------------------------------
    {
        GcmWorker w(a);
        std::vector<unsigned char> scratch(kIvBytes + 16 * static_cast<size_t>(dim) + kTagBytes + 64);
        std::vector<double> row(dim);
        t0 = now();
        long ok = 0;
        for (int64_t i = 0; i < n; i++) ok += pointstore_open_one(&ps, w, order[i], scratch, row.data(), nullptr);
        const double dt = now() - t0;
        printf("pointstore_open_one, 1 thread, scattered: %.3f us per record (%.2f GB/s), ok %ld/%lld\n", dt / n * 1e6, (8.0 * dim + 16) * n / dt / 1e9, ok, (long long)n);
    }
    // ---- the pieces of the store's open around the cipher -------------------------------------------------------------------------------
    {
        std::vector<unsigned char> scratch(kIvBytes + 16 * static_cast<size_t>(dim) + kTagBytes + 64);
        t0 = now();
        unsigned long acc = 0;
        for (int64_t i = 0; i < n; i++) { copy_from_shared(scratch.data(), ps.rec(order[i]) + kRecHeader, kIvBytes + 8 * dim + kTagBytes); acc += scratch[5]; }
        printf("  snapshot copy, scattered    %.3f us (%lu)\n", (now() - t0) / n * 1e6, acc);
        t0 = now();
        for (int64_t i = 0; i < n; i++) {
            if (i + 1 < n) { const unsigned char* r = ps.rec(order[i + 1]); for (size_t off = 0; off < ps.stride; off += 64) __builtin_prefetch(r + off, 0, 0); }
            copy_from_shared(scratch.data(), ps.rec(order[i]) + kRecHeader, kIvBytes + 8 * dim + kTagBytes); acc += scratch[5];
        }
        printf("  snapshot copy + prefetch    %.3f us (%lu)\n", (now() - t0) / n * 1e6, acc);
        GcmWorker w(a);
        unsigned char key[32];
        ps.key_for(1, key);
        w.set_dec_key(1, key);
        char aad[96];
        std::vector<float> row(dim);
        t0 = now();
        long ok = 0;
        for (int64_t i = 0; i < n; i++) {      // no snapshot: the cipher reads the record in place (what a store without live writers could do)
            const int64_t h = order[i];
            const unsigned char* r = ps.rec(h) + kRecHeader;
            const int al = aad_for(aad, sizeof(aad), h, 1, dim);
            unsigned char* pt = scratch.data();
            ok += w.open(r, reinterpret_cast<const unsigned char*>(aad), al, r + kIvBytes, 8 * dim, pt);
            decode_row_fast(pt, dim, row.data());
        }
        printf("  in-place open + decode      %.3f us (ok %ld)\n", (now() - t0) / n * 1e6, ok);
    }
    // ---- the batch call as the pipeline makes it ---------------------------------------------------------------------------------
    {
        const int64_t B = 256, nq = n / B;
        std::vector<int32_t> cnt(nq, static_cast<int32_t>(B)), oid(n), ocnt(nq);
        std::vector<float> dst(static_cast<size_t>(n) * dim);
        for (int rep = 0; rep < 2; rep++) {
            t0 = now();
            pointstore_open_batch<float>(&ps, nq, B, order.data(), cnt.data(), dst.data(), oid.data(), ocnt.data(), threads);
            const double dt = now() - t0;
            printf("pointstore_open_batch<float>, %d threads: %.1f ms per %lld records = %.3f us per record per thread, %.2f M opens/s\n", threads, dt * 1e3, (long long)n,
                   dt / n * 1e6 * threads, n / dt / 1e6);
        }
        long bad = 0;
        for (int64_t i = 0; i < n; i++) for (int j = 0; j < dim; j += 37) bad += dst[i * dim + j] != X[static_cast<size_t>(order[i]) * dim + j];
        printf("rows equal the plaintext: %s\n", bad ? "NO" : "yes");
    }
    return 0;
}
