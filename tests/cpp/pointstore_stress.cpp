// CPU stress of the product's point store (host/pointstore.hpp) — config #5's situation without a GPU: reader threads open
// batches of records (QSI stage B's host half) while a writer rotates the key and migrates every record (KeyRotationServiceImpl
// reencryptTouched), and another re-seals records under the CURRENT version (encryptToPoint of an existing id).  Every open must
// succeed and return the plaintext that was sealed.  Built by tests/test_hostpipe_sanitizers.py with -fsanitize=thread and with
// -fsanitize=address,undefined: the reader / writer protocol on the per-record version word is checked by a race detector here,
// because GPU sanitizers do not exist on the pool.  Exit code 0 = no failed open, no wrong value.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../fspann-query-system_amd/host/pointstore.hpp"

using namespace fspann;

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 4000;
    const int dim = argc > 2 ? atoi(argv[2]) : 24;
    const int rounds = argc > 3 ? atoi(argv[3]) : 3;
    if (!crypto_api()) { std::printf("libcrypto not found: skipped\n"); return 77; }
    fspann_pointstore ps;
    ps.n = n; ps.dim = dim;
    ps.stride = pointstore_stride(dim);
    ps.mem.assign(static_cast<size_t>(n) * ps.stride, 0);
    for (int i = 0; i < 32; i++) ps.master[i] = static_cast<unsigned char>(7 * i + 1);
    ps.have_master = true;
    std::vector<float> X(static_cast<size_t>(n) * dim);
    for (size_t i = 0; i < X.size(); i++) X[i] = static_cast<float>((i * 2654435761u) % 100003) * 0.25f - 7.0f;
    std::atomic<long long> bad{0};
    if (pointstore_encrypt<float>(&ps, 0, n, X.data(), 4, &bad) != 0 || bad.load()) { std::printf("initial encrypt failed\n"); return 1; }

    std::atomic<bool> stop{false};
    std::atomic<long long> wrong{0}, failed{0}, opened{0};
    const int64_t nq = 16, B = 64;
    auto reader = [&](unsigned seed) {
        std::vector<int32_t> ids(nq * B), cnt(nq, static_cast<int32_t>(B)), oid(nq * B), ocnt(nq);
        std::vector<float> dst(static_cast<size_t>(nq) * B * dim);
        unsigned s = seed;
        while (!stop.load(std::memory_order_relaxed)) {
            for (auto& id : ids) { s = s * 1664525u + 1013904223u; id = static_cast<int32_t>((s >> 8) % n); }
            pointstore_open_batch<float>(&ps, nq, B, ids.data(), cnt.data(), dst.data(), oid.data(), ocnt.data(), 2);
            for (int64_t q = 0; q < nq; q++) {
                if (ocnt[q] != B) failed += B - ocnt[q];
                for (int j = 0; j < ocnt[q]; j++) {
                    const int32_t id = oid[q * B + j];
                    const float* got = dst.data() + (q * B + j) * dim;
                    for (int t = 0; t < dim; t++)
                        if (got[t] != X[static_cast<size_t>(id) * dim + t]) { wrong++; break; }
                }
                opened += ocnt[q];
            }
        }
    };
    std::vector<std::thread> readers;
    for (int r = 0; r < 3; r++) readers.emplace_back(reader, 1234u + 77u * r);
    std::thread resealer([&] {       // re-seal under the CURRENT version: v -> -1 -> v, the case the version word alone cannot see
        unsigned s = 99;
        std::atomic<long long> b2{0};
        while (!stop.load(std::memory_order_relaxed)) {
            s = s * 1664525u + 1013904223u;
            const int64_t h0 = (s >> 8) % (n - 8);
            pointstore_encrypt<float>(&ps, h0, 8, X.data() + static_cast<size_t>(h0) * dim, 1, &b2);
        }
        if (b2.load()) wrong += b2.load();
    });
    std::vector<int32_t> all(n);
    for (int64_t i = 0; i < n; i++) all[i] = static_cast<int32_t>(i);
    long long migrated = 0;
    for (int r = 0; r < rounds; r++) {
        ps.current_version.fetch_add(1);          // rotateKeyOnly
        long long done = 0;
        if (pointstore_reencrypt(&ps, all.data(), n, 3, &done) != 0) { std::printf("reencrypt failed\n"); stop = true; break; }
        migrated += done;
    }
    stop = true;
    for (auto& t : readers) t.join();
    resealer.join();
    std::printf("opened %lld failed %lld wrong %lld migrated %lld\n", opened.load(), failed.load(), wrong.load(), migrated);
    return (failed.load() == 0 && wrong.load() == 0 && opened.load() > 0 && migrated > 0) ? 0 : 1;
}
