// host_mirror_test.cpp — drives the C++ operator mirror (tests/cpp/fspann_host.hpp: test infrastructure, the tested twin of the product is operators.py) the way
// ForwardSecureANNSystem does (FSA:479-570 batchInsert, :622-748 runQueries): insert -> finalizeForSearch ->
// createToken -> search, and dumps results for tests/test_gpu_cpp_host.py to compare with the golden fixtures.
// Also checks the error behaviour of the operator surface (it/.../SuperFailureModeIT.java:11-46).
//
// usage: host_mirror_test <in.bin> <out.bin>
//   in : int64 header {n, d, T, D, m, lam, B, K, seed, nq, hard_cap, preinstall}; X[n*d] f64; Q[nq*d] f64;
//        if preinstall: alpha[T*D*m*d], r[T*D*m], omega[T*D*m] f64
//   out: per query: int32 count, int32 ids[K] (decimal id parsed back), f64 dist[K], int32 metrics[4]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "fspann_host.hpp"

using namespace fspann::host;

// plaintext stand-in for AesGcmCryptoService + KeyRotationServiceImpl + RocksDBMetadataManager
struct PlainHost : CryptoService, KeyLifeCycleService, MetadataManager {
    std::map<std::string, EncryptedPoint> points;
    std::set<std::string> deleted;
    static std::vector<uint8_t> enc(const std::vector<double>& v) {
        std::vector<uint8_t> b(v.size() * 8);
        std::memcpy(b.data(), v.data(), b.size());
        return b;
    }
    static std::vector<double> dec(const std::vector<uint8_t>& b) {
        std::vector<double> v(b.size() / 8);
        std::memcpy(v.data(), b.data(), b.size());
        return v;
    }
    KeyVersion getCurrentVersion() override { return {1, std::vector<uint8_t>(32, 0)}; }
    KeyVersion getVersion(int v) override { return {v, std::vector<uint8_t>(32, 0)}; }
    EncryptedPoint encrypt(const std::string& id, const std::vector<double>& v, const KeyVersion& kv) override {
        EncryptedPoint ep; ep.id = id; ep.version = kv.version; ep.iv.assign(12, 0); ep.ciphertext = enc(v); ep.dim = (int)v.size();
        return ep;
    }
    std::vector<double> decryptFromPoint(const EncryptedPoint& ep, const std::vector<uint8_t>&) override { return dec(ep.ciphertext); }
    std::vector<uint8_t> encryptQuery(const std::vector<double>& v, const std::vector<uint8_t>&, const std::vector<uint8_t>&) override { return enc(v); }
    std::vector<double> decryptQuery(const std::vector<uint8_t>& ct, const std::vector<uint8_t>&, const std::vector<uint8_t>&) override { return dec(ct); }
    bool isDeleted(const std::string& id) override { return deleted.count(id) != 0; }
    void saveEncryptedPoint(const EncryptedPoint& ep) override { points[ep.id] = ep; }
    bool loadEncryptedPoint(const std::string& id, EncryptedPoint* out) override {
        auto it = points.find(id);
        if (it == points.end()) return false;
        *out = it->second;
        return true;
    }
};

#define EXPECT_THROW(stmt, EXC)                                                          \
    do {                                                                                 \
        bool ok_ = false;                                                                \
        try { stmt; } catch (const EXC&) { ok_ = true; } catch (...) {}                  \
        if (!ok_) { std::fprintf(stderr, "FAIL: %s did not throw %s\n", #stmt, #EXC); return 2; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 3) return 64;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 65;
    int64_t h[12];
    if (std::fread(h, 8, 12, f) != 12) return 66;
    const int64_t n = h[0], d = h[1], T = h[2], D = h[3], m = h[4], lam = h[5], B = h[6], K = h[7], seed = h[8], nq = h[9], hard_cap = h[10], pre = h[11];
    std::vector<double> X(n * d), Q(nq * d);
    if (std::fread(X.data(), 8, X.size(), f) != X.size() || std::fread(Q.data(), 8, Q.size(), f) != Q.size()) return 67;
    SystemConfig cfg;
    cfg.m = (int)m; cfg.lambda = (int)lam; cfg.divisions = (int)D; cfg.tables = (int)T; cfg.seed = seed;
    cfg.refinementLimit = (int)B; cfg.maxGlobalCandidates = (int)hard_cap; cfg.kVariants = {(int)K};
    GFunctionRegistry::reset();
    if (pre) {
        const size_t P = (size_t)(T * D * m);
        std::vector<double> a(P * d), r(P), w(P);
        if (std::fread(a.data(), 8, a.size(), f) != a.size() || std::fread(r.data(), 8, P, f) != P || std::fread(w.data(), 8, P, f) != P) return 68;
        GFunctionRegistry::install(a, r, w, (int)d, (int)m, (int)lam, seed, (int)T, (int)D);
    }
    std::fclose(f);

    PlainHost host;
    PartitionedIndexService index(&host, &cfg, &host, &host, 0);
    QueryTokenFactory tf(&host, &host, &cfg, &index);
    QueryServiceImpl qs(&index, &host, &host, &tf, &cfg);

    // ---- error behaviour before the index exists -----------------------------------------------------
    std::vector<double> q0(Q.begin(), Q.begin() + d);
    EXPECT_THROW(PartitionedIndexService(nullptr, &cfg, &host, &host), NullPointerException);
    EXPECT_THROW(index.insert(nullptr, &q0), NullPointerException);
    EXPECT_THROW(tf.create(nullptr, 5), NullPointerException);
    EXPECT_THROW(tf.create(q0, 0), IllegalArgumentException);
    if (!pre) EXPECT_THROW(tf.create(q0, 5), IllegalStateException);                 // registry not initialised
    {
        QueryToken t0(std::vector<uint64_t>((size_t)(T * D * ((m * lam + 63) / 64)), 0), {}, {}, 5, (int)T, (int)d, 1, (int)lam, "x");
        EXPECT_THROW(index.lookupCandidatesWithScores(&t0), IllegalStateException);   // "Index not finalized" (PIS:594)
        EXPECT_THROW(index.lookupCandidatesWithScores(nullptr), NullPointerException);
    }
    if (qs.search(nullptr).size() != 0) return 3;                                      // null token -> empty (QSI:102)

    for (int64_t i = 0; i < n; i++) index.insert(std::to_string(i), std::vector<double>(X.begin() + i * d, X.begin() + (i + 1) * d));
    index.finalizeForSearch();
    index.finalizeForSearch();                                                          // idempotent (PIS:790-793)
    if (!index.isFrozen()) return 4;
    {
        std::vector<double> bad = q0;
        bad[0] = std::nan("");
        EXPECT_THROW(tf.create(bad, 5), IllegalArgumentException);                      // "Vector contains NaN/Inf"
        std::vector<double> wrong(d + 1, 0.0);
        EXPECT_THROW(tf.create(wrong, 5), IllegalStateException);                       // dimension mismatch vs registry
        EXPECT_THROW(index.insert(std::to_string(n), wrong), IllegalArgumentException); // mixed dimensions
    }

    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 69;
    for (int64_t qi = 0; qi < nq; qi++) {
        QueryToken tok = tf.create(std::vector<double>(Q.begin() + qi * d, Q.begin() + (qi + 1) * d), (int)K);
        std::vector<QueryResult> res = qs.search(&tok);
        int32_t cnt = (int32_t)res.size();
        std::vector<int32_t> ids((size_t)K, -1);
        std::vector<double> dist((size_t)K, INFINITY);
        for (int i = 0; i < cnt; i++) { ids[i] = std::atoi(res[i].id.c_str()); dist[i] = res[i].distance; }
        int32_t met[4] = {qs.getLastCandTotal(), qs.getLastCandKept(), qs.getLastCandDecrypted(), qs.getLastReturned()};
        std::fwrite(&cnt, 4, 1, o);
        std::fwrite(ids.data(), 4, (size_t)K, o);
        std::fwrite(dist.data(), 8, (size_t)K, o);
        std::fwrite(met, 4, 4, o);
        // derive keeps the codes, changes topK
        QueryToken t2 = tf.derive(&tok, 3);
        if (t2.getTopK() != 3 || t2.getBitCodes() != tok.getBitCodes()) return 5;
    }
    std::fclose(o);
    GFunctionRegistry::reset();
    std::printf("host mirror ok: %lld queries\n", (long long)nq);
    return 0;
}
