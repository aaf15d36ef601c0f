// fspann_host.hpp — C++17 host-side mirror of the reference's TokenGen / Route / Refine operator
// surface, above the C ABI (include/fspann.h).  Header-only; link with -lfspann_hip.
//
// The reference is Java (compiled code) and no JDK exists in the build container, so the host side
// above the ABI is written in C++ with the reference's class and method names, argument meaning and
// error behaviour:
//
//   QueryTokenFactory::create / derive       qry/core/QueryTokenFactory.java:63,182
//   PartitionedIndexService::*               idx/PartitionedIndexService.java:265-347,459-896
//   QueryServiceImpl::search + getLast*      qry/service/QueryServiceImpl.java:101-352,417-474
//   GFunctionRegistry (process-wide static)  idx/GFunctionRegistry.java:63-252
//
// Java exception -> C++: IllegalStateException / IllegalArgumentException / NullPointerException below.
// AES-GCM, key versions and point storage stay on the host behind the same three collaborators the
// reference wires in (CryptoService, KeyLifeCycleService, RocksDBMetadataManager) — abstract here.
// The JVM twin is java/com/fspann/gpu + jni/; the Python twin is operators.py.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/fspann.h"

namespace fspann::host {

struct IllegalStateException : std::logic_error { using std::logic_error::logic_error; };
struct IllegalArgumentException : std::invalid_argument { using std::invalid_argument::invalid_argument; };
struct NullPointerException : std::logic_error { using std::logic_error::logic_error; };
struct DeviceException : std::runtime_error { using std::runtime_error::runtime_error; };

inline void check(int rc) {
    if (rc == FSPANN_OK) return;
    const std::string msg = fspann_last_error();
    switch (rc) {
        case FSPANN_E_STATE: throw IllegalStateException(msg);
        case FSPANN_E_ARG: throw IllegalArgumentException(msg);
        case FSPANN_E_NULL: throw NullPointerException(msg);
        default: throw DeviceException("fspann(" + std::to_string(rc) + "): " + msg);
    }
}

// ---- value types (common/QueryResult.java, QueryToken.java, EncryptedPoint.java) ------------------
struct QueryResult { std::string id; double distance; };
struct CandidateWithScore { std::string id; long hammingDist; };   // PIS:82-89
struct KeyVersion { int version; std::vector<uint8_t> key; };
struct EncryptedPoint { std::string id; int version = 0; std::vector<uint8_t> iv, ciphertext; int dim = 0; };

// config/SystemConfig.java:237-338 — the knobs the path reads
struct SystemConfig {
    int m = 24, lambda = 2, divisions = 3, tables = 6;
    int64_t seed = 13;
    int refinementLimit = 20000, maxGlobalCandidates = 20000, probeOverride = -1, hammingPrefilterThreshold = 0;
    std::vector<int> kVariants{1, 10, 20, 40, 60, 80, 100};
    fspann_cfg native(int dim) const {
        return fspann_cfg{tables, divisions, m, lambda, dim, 64, 5, probeOverride, maxGlobalCandidates, refinementLimit,
                          hammingPrefilterThreshold, 0};
    }
};

// ---- host collaborators (unchanged subsystems of the reference) -------------------------------------
struct KeyLifeCycleService {
    virtual ~KeyLifeCycleService() = default;
    virtual KeyVersion getCurrentVersion() = 0;
    virtual KeyVersion getVersion(int v) = 0;
};
struct CryptoService {
    virtual ~CryptoService() = default;
    virtual EncryptedPoint encrypt(const std::string& id, const std::vector<double>& v, const KeyVersion& kv) = 0;
    virtual std::vector<double> decryptFromPoint(const EncryptedPoint& ep, const std::vector<uint8_t>& key) = 0;
    virtual std::vector<uint8_t> encryptQuery(const std::vector<double>& v, const std::vector<uint8_t>& key,
                                              const std::vector<uint8_t>& iv) = 0;
    virtual std::vector<double> decryptQuery(const std::vector<uint8_t>& ct, const std::vector<uint8_t>& iv,
                                             const std::vector<uint8_t>& key) = 0;
};
struct MetadataManager {   // RocksDBMetadataManager: isDeleted / saveEncryptedPoint / loadEncryptedPoint
    virtual ~MetadataManager() = default;
    virtual bool isDeleted(const std::string& id) = 0;
    virtual void saveEncryptedPoint(const EncryptedPoint& ep) = 0;
    virtual bool loadEncryptedPoint(const std::string& id, EncryptedPoint* out) = 0;
};

// RAII handle of one fspann_ctx
class Context {
  public:
    Context(const SystemConfig& cfg, int dim, int device = 0) {
        fspann_cfg c = cfg.native(dim);
        check(fspann_ctx_create(device, &c, &ctx_));
    }
    ~Context() { fspann_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    fspann_ctx* get() const { return ctx_; }
  private:
    fspann_ctx* ctx_ = nullptr;
};

inline int32_t javaStringHash(const std::string& s) {   // String.hashCode (ASCII ids)
    uint32_t h = 0;
    for (unsigned char c : s) h = 31u * h + c;
    return static_cast<int32_t>(h);
}

// ---- GFunctionRegistry: process-wide static, like the reference ----------------------------------------
class GFunctionRegistry {
  public:
    struct State {
        bool initialized = false;
        int DIM = -1, M = -1, LAMBDA = -1, TABLES = -1, DIVISIONS = -1;
        int64_t BASE_SEED = -1;
        std::vector<double> alpha, r, omega;
    };
    static State& st() { static State s; return s; }
    static bool isInitialized() { return st().initialized; }
    static void reset() { st() = State(); }
    // idx/GFunctionRegistry.java:63-147; the projection pass of buildFromSample runs on `ctx`'s GPU
    static void initialize(const std::vector<std::vector<double>>& sample, int dimension, int m, int lambda, int64_t baseSeed,
                           int tables, int divisions, fspann_ctx* ctx) {
        if (sample.empty()) throw IllegalArgumentException("Sample vectors cannot be empty");
        for (auto& v : sample)
            if (static_cast<int>(v.size()) != dimension)
                throw IllegalArgumentException("Mixed dimensions in GFunctionRegistry sample: expected " + std::to_string(dimension));
        State& s = st();
        if (s.initialized && s.DIM == dimension && s.M == m && s.LAMBDA == lambda && s.BASE_SEED == baseSeed &&
            s.TABLES == tables && s.DIVISIONS == divisions)
            return;   // :86-95 same configuration -> no-op
        std::vector<double> flat;
        flat.reserve(sample.size() * dimension);
        for (auto& v : sample) flat.insert(flat.end(), v.begin(), v.end());
        check(fspann_registry_initialize(ctx, flat.data(), static_cast<int64_t>(sample.size()), baseSeed));
        const size_t P = static_cast<size_t>(tables) * divisions * m;
        std::vector<double> a(P * dimension), r(P), w(P);
        check(fspann_get_gfunctions(ctx, a.data(), r.data(), w.data()));
        install(a, r, w, dimension, m, lambda, baseSeed, tables, divisions);
    }
    // import GFunctions generated elsewhere (e.g. exported from the JVM)
    static void install(std::vector<double> alpha, std::vector<double> r, std::vector<double> omega, int dimension, int m,
                        int lambda, int64_t baseSeed, int tables, int divisions) {
        for (double w : omega)
            if (!(w > 0.0)) throw IllegalArgumentException("omega_j <= 0");
        State& s = st();
        s.alpha = std::move(alpha); s.r = std::move(r); s.omega = std::move(omega);
        s.DIM = dimension; s.M = m; s.LAMBDA = lambda; s.BASE_SEED = baseSeed; s.TABLES = tables; s.DIVISIONS = divisions;
        s.initialized = true;
    }
};

// ---- QueryToken (common/QueryToken.java:49-71): bitCodes = uint64[T][D][W] BitSet words -------------------
class QueryToken {
  public:
    QueryToken(std::vector<uint64_t> bitCodes, std::vector<uint8_t> iv, std::vector<uint8_t> ct, int topK, int numTables,
               int dimension, int version, int lambda, std::string ctx)
        : bitCodes_(std::move(bitCodes)), iv_(std::move(iv)), ct_(std::move(ct)), topK_(std::max(1, topK)),
          numTables_(std::max(1, numTables)), dimension_(dimension), version_(version), lambda_(lambda), ctx_(std::move(ctx)) {}
    const std::vector<uint64_t>& getBitCodes() const { return bitCodes_; }
    const std::vector<uint8_t>& getIv() const { return iv_; }
    const std::vector<uint8_t>& getEncryptedQuery() const { return ct_; }
    int getTopK() const { return topK_; }
    int getNumTables() const { return numTables_; }
    int getDimension() const { return dimension_; }
    int getVersion() const { return version_; }
    int getLambda() const { return lambda_; }
    const std::string& getEncryptionContext() const { return ctx_; }
  private:
    std::vector<uint64_t> bitCodes_;
    std::vector<uint8_t> iv_, ct_;
    int topK_, numTables_, dimension_, version_, lambda_;
    std::string ctx_;
};

// ---- PartitionedIndexService: Setup + Route ---------------------------------------------------------------
class PartitionedIndexService {
  public:
    static constexpr int MIN_SAMPLE_SIZE = 1000, MAX_SAMPLE_SIZE = 10000, DEFAULT_MAX_PROBES = 5;   // PIS:50-51,93

    PartitionedIndexService(MetadataManager* metadata, const SystemConfig* cfg, KeyLifeCycleService* keyService,
                            CryptoService* cryptoService, int device = 0)
        : metadata_(metadata), cfg_(cfg), keys_(keyService), crypto_(cryptoService), device_(device) {
        if (!metadata) throw NullPointerException("metadata");
        if (!cfg) throw NullPointerException("cfg");
        if (!keyService) throw NullPointerException("keyService");
        if (!cryptoService) throw NullPointerException("cryptoService");
    }

    void insert(const std::string* id, const std::vector<double>* vector) {   // PIS:265-312
        if (!id) throw NullPointerException("id cannot be null");
        if (!vector) throw NullPointerException("vector cannot be null");
        if (GFunctionRegistry::isInitialized()) {
            if (static_cast<int>(vector->size()) != GFunctionRegistry::st().DIM)
                throw IllegalArgumentException("Mixed dimensions not supported in single index: got " +
                                               std::to_string(vector->size()) + ", expected " + std::to_string(GFunctionRegistry::st().DIM));
        } else {
            if (static_cast<int>(sample_.size()) < MAX_SAMPLE_SIZE) sample_.push_back(*vector);
            if (static_cast<int>(sample_.size()) >= MIN_SAMPLE_SIZE) initializeRegistry();
        }
        if (!GFunctionRegistry::isInitialized()) {   // stage plaintext for later indexing (PIS:292-298)
            pending_.emplace_back(*id, *vector);
            return;
        }
        stage(crypto_->encrypt(*id, *vector, keys_->getCurrentVersion()), *vector);
    }
    void insert(const std::string& id, const std::vector<double>& v) { insert(&id, &v); }

    void finalizeForSearch() {   // PIS:789-845
        if (frozen_) return;
        if (!GFunctionRegistry::isInitialized()) {
            if (static_cast<int>(sample_.size()) >= MIN_SAMPLE_SIZE) initializeRegistry();
            else throw IllegalStateException("Cannot finalize index: only " + std::to_string(sample_.size()) + " samples collected (< MIN_SAMPLE_SIZE)");
        }
        const auto& g = GFunctionRegistry::st();
        if (g.M != cfg_->m || g.LAMBDA != cfg_->lambda || g.TABLES != cfg_->tables || g.DIVISIONS != cfg_->divisions)
            throw IllegalStateException("GFunctionRegistry mismatch at finalize");
        for (auto& pv : pending_) stage(crypto_->encrypt(pv.first, pv.second, keys_->getCurrentVersion()), pv.second);
        pending_.clear();
        if (!ids_.empty()) {
            ensureCtx(static_cast<int>(stagedVecs_.size() / ids_.size()));
            check(fspann_set_gfunctions(ctx_->get(), g.alpha.data(), g.r.data(), g.omega.data()));
            pushIdMeta();
            // handles were assigned in staged order: order == identity over the staged list
            const std::vector<int32_t> ord = identity();
            check(fspann_build_index(ctx_->get(), static_cast<int64_t>(ids_.size()), stagedVecs_.data(), FSPANN_F64, ord.data()));
            stagedVecs_.clear();
            stagedVecs_.shrink_to_fit();
        }
        frozen_ = true;
    }

    // PIS:592-715
    std::vector<CandidateWithScore> lookupCandidatesWithScores(const QueryToken* token) {
        std::vector<int32_t> ids, score;
        if (!route(token, INT32_MAX, &ids, &score, nullptr)) return {};
        std::vector<CandidateWithScore> out;
        out.reserve(ids.size());
        lastTouched_.clear();
        for (size_t i = 0; i < ids.size(); i++) { out.push_back({ids_[ids[i]], score[i]}); lastTouched_.push_back(ids_[ids[i]]); }
        return out;
    }
    // PIS:459-582 (list truncated to HARD_CAP, :558-565)
    std::vector<std::string> lookupCandidateIds(const QueryToken* token) {
        std::vector<int32_t> ids;
        if (!route(token, std::max(cfg_->maxGlobalCandidates, cfg_->refinementLimit), &ids, nullptr, nullptr)) return {};
        lastTouched_.clear();
        for (int32_t h : ids) lastTouched_.push_back(ids_[h]);
        return lastTouched_;
    }
    bool loadPointIfActive(const std::string& id, EncryptedPoint* out) {   // PIS:717-724
        if (metadata_->isDeleted(id)) return false;
        try { return metadata_->loadEncryptedPoint(id, out); } catch (...) { return false; }
    }
    bool isFrozen() const { return frozen_; }
    int numTables() const { return cfg_->tables; }
    int getDefaultMaxProbes() const { return DEFAULT_MAX_PROBES; }
    void setProbeOverride(int probes) { probeOverride_ = probes; }
    void clearProbeOverride() { probeOverride_ = -1; }
    int getLastRawCandidateCount() const { return lastRaw_; }
    int getLastTouchedCount() const { return static_cast<int>(lastTouched_.size()); }
    const std::vector<std::string>& getLastTouchedIds() const { return lastTouched_; }
    fspann_ctx* nativeContext() { return ctx_ ? ctx_->get() : nullptr; }
    int dimension() const { return dim_; }

    // Route for QueryServiceImpl: first `limit` entries of the reference's list; returns false for "no such dim".
    bool route(const QueryToken* token, int limit, std::vector<int32_t>* ids, std::vector<int32_t>* score, int* kept) {
        if (!token) throw NullPointerException("token");
        if (!frozen_) throw IllegalStateException("Index not finalized");                       // PIS:594
        if (!ctx_ || token->getDimension() != dim_) return false;                               // PIS:598
        const int TD = cfg_->tables * cfg_->divisions, W = (cfg_->m * cfg_->lambda + 63) / 64;
        if (token->getBitCodes().empty()) throw IllegalStateException("MSANNP violation: QueryToken missing BitSet codes");
        if (static_cast<int>(token->getBitCodes().size()) != TD * W)
            throw IllegalStateException("Token tables mismatch: token=" + std::to_string(token->getNumTables()) + " index=" + std::to_string(cfg_->tables));
        const int64_t cap = std::max<int64_t>(1, std::min<int64_t>(limit, fspann_route_max_candidates(ctx_->get(), probeOverride_)));
        ids->assign(static_cast<size_t>(cap), -1);
        std::vector<int32_t> sc(static_cast<size_t>(cap), -1);
        int32_t count = 0, k = 0, raw = 0;
        check(fspann_route(ctx_->get(), 1, token->getBitCodes().data(), probeOverride_, limit, cap, ids->data(), sc.data(), &count, &k, &raw));
        ids->resize(count);
        sc.resize(count);
        if (score) *score = std::move(sc);
        if (kept) *kept = k;
        lastRaw_ = raw;
        return true;
    }
    const std::string& idOf(int32_t handle) const { return ids_[handle]; }

  private:
    void ensureCtx(int dim) {
        if (!ctx_) { dim_ = dim; ctx_ = std::make_unique<Context>(*cfg_, dim, device_); }
    }
    void initializeRegistry() {   // PIS:161-245
        if (static_cast<int>(sample_.size()) < MIN_SAMPLE_SIZE)
            throw IllegalStateException("Refusing to initialize GFunctionRegistry with sampleSize=" + std::to_string(sample_.size()));
        const int dim = static_cast<int>(sample_[0].size());
        ensureCtx(dim);
        GFunctionRegistry::initialize(sample_, dim, cfg_->m, cfg_->lambda, cfg_->seed, cfg_->tables, cfg_->divisions, ctx_->get());
        sample_.clear();
    }
    void stage(const EncryptedPoint& ep, const std::vector<double>& vec) {   // PIS:314-347 (codes are computed in bulk at finalize)
        metadata_->saveEncryptedPoint(ep);
        auto it = handle_.find(ep.id);
        if (it != handle_.end()) {   // HashMap.put of an existing key: position kept, code replaced
            std::copy(vec.begin(), vec.end(), stagedVecs_.begin() + static_cast<size_t>(it->second) * vec.size());
            return;
        }
        handle_[ep.id] = static_cast<int32_t>(ids_.size());
        ids_.push_back(ep.id);
        stagedVecs_.insert(stagedVecs_.end(), vec.begin(), vec.end());
    }
    void pushIdMeta() {
        std::vector<int32_t> jh(ids_.size());
        std::vector<uint8_t> del(ids_.size());
        for (size_t i = 0; i < ids_.size(); i++) { jh[i] = javaStringHash(ids_[i]); del[i] = metadata_->isDeleted(ids_[i]) ? 1 : 0; }
        check(fspann_set_id_meta(ctx_->get(), static_cast<int64_t>(ids_.size()), jh.data(), del.data()));
    }
    std::vector<int32_t> identity() const {
        std::vector<int32_t> o(ids_.size());
        for (size_t i = 0; i < o.size(); i++) o[i] = static_cast<int32_t>(i);
        return o;
    }

    MetadataManager* metadata_;
    const SystemConfig* cfg_;
    KeyLifeCycleService* keys_;
    CryptoService* crypto_;
    int device_;
    std::unique_ptr<Context> ctx_;
    int dim_ = -1;
    bool frozen_ = false;
    std::vector<std::vector<double>> sample_;
    std::vector<std::pair<std::string, std::vector<double>>> pending_;
    std::vector<std::string> ids_;
    std::unordered_map<std::string, int32_t> handle_;
    std::vector<double> stagedVecs_;
    int probeOverride_ = -1, lastRaw_ = 0;
    std::vector<std::string> lastTouched_;
};

// ---- QueryTokenFactory: TokenGen ----------------------------------------------------------------------------
class QueryTokenFactory {
  public:
    QueryTokenFactory(CryptoService* crypto, KeyLifeCycleService* keyService, const SystemConfig* cfg, PartitionedIndexService* index)
        : crypto_(crypto), keys_(keyService), cfg_(cfg), index_(index) {
        if (!crypto || !keyService || !cfg) throw NullPointerException("QueryTokenFactory dependency");
    }
    QueryToken create(const std::vector<double>* vec, int topK) {   // QueryTokenFactory.java:63-167
        if (!vec) throw NullPointerException("query vector is null");
        if (topK <= 0) throw IllegalArgumentException("topK must be > 0");
        if (!GFunctionRegistry::isInitialized()) throw IllegalStateException("GFunctionRegistry not initialized. Build index first.");
        const auto& g = GFunctionRegistry::st();
        const int dim = static_cast<int>(vec->size());
        if (g.DIM != dim || g.TABLES != cfg_->tables || g.DIVISIONS != cfg_->divisions || g.M != cfg_->m || g.LAMBDA != cfg_->lambda)
            throw IllegalStateException("GFunctionRegistry mismatch");
        fspann_ctx* ctx = index_ ? index_->nativeContext() : nullptr;
        if (!ctx) throw IllegalStateException("GFunctionRegistry not initialized. Build index first.");
        const int TD = cfg_->tables * cfg_->divisions, W = (cfg_->m * cfg_->lambda + 63) / 64;
        std::vector<uint64_t> codes(static_cast<size_t>(TD) * W);
        check(fspann_encode(ctx, 1, vec->data(), FSPANN_F64, codes.data(), nullptr));   // NaN/Inf -> IllegalArgumentException
        KeyVersion kv = keys_->getCurrentVersion();
        std::vector<uint8_t> iv(12);
        for (auto& b : iv) b = static_cast<uint8_t>(rand());
        return QueryToken(std::move(codes), iv, crypto_->encryptQuery(*vec, kv.key, iv), topK, cfg_->tables, dim, kv.version,
                          cfg_->lambda, "dim_" + std::to_string(dim) + "_v" + std::to_string(kv.version));
    }
    QueryToken create(const std::vector<double>& vec, int topK) { return create(&vec, topK); }
    QueryToken derive(const QueryToken* tok, int newTopK) {   // :182-198
        if (!tok) throw NullPointerException("token is null");
        if (newTopK <= 0) throw IllegalArgumentException("newTopK must be > 0");
        return QueryToken(tok->getBitCodes(), tok->getIv(), tok->getEncryptedQuery(), newTopK, tok->getNumTables(), tok->getDimension(),
                          tok->getVersion(), tok->getLambda(), tok->getEncryptionContext());
    }
  private:
    CryptoService* crypto_;
    KeyLifeCycleService* keys_;
    const SystemConfig* cfg_;
    PartitionedIndexService* index_;
};

// ---- QueryServiceImpl: Refine ---------------------------------------------------------------------------------
class QueryServiceImpl {
  public:
    QueryServiceImpl(PartitionedIndexService* index, CryptoService* crypto, KeyLifeCycleService* keyService, QueryTokenFactory* tf,
                     const SystemConfig* cfg)
        : index_(index), crypto_(crypto), keys_(keyService), tf_(tf), cfg_(cfg) {
        if (!index) throw NullPointerException("index");
        if (!crypto) throw NullPointerException("cryptoService");
        if (!keyService) throw NullPointerException("keyService");
        if (!cfg) throw NullPointerException("cfg");
    }

    std::vector<QueryResult> search(const QueryToken* token) {   // QSI:101-352
        if (!token) return {};
        lastCandTotal_ = lastCandKept_ = lastCandDecrypted_ = lastReturned_ = 0;
        lastCandIds_.clear();
        touched_.clear();
        KeyVersion qkv;
        try { qkv = keys_->getVersion(token->getVersion()); } catch (...) { qkv = keys_->getCurrentVersion(); }
        const std::vector<double> q = crypto_->decryptQuery(token->getEncryptedQuery(), token->getIv(), qkv.key);
        for (double x : q) if (!std::isfinite(x)) return {};   // :137-140
        const int K = token->getTopK();
        bool retried = false;
        struct Guard { PartitionedIndexService* i; ~Guard() { i->clearProbeOverride(); } } guard{index_};   // finally (:342-343)
        while (true) {
            const int runtimeLimit = getEffectiveRefinementLimit(cfg_->refinementLimit);
            std::vector<int32_t> sel;
            int kept = 0;
            if (!index_->route(token, runtimeLimit, &sel, nullptr, &kept)) return {};   // stage A + A.5 on the GPU
            lastCandTotal_ = index_->getLastRawCandidateCount();
            lastCandKept_ = kept;
            if (kept == 0) return {};
            lastUnique_ = static_cast<int>(sel.size());
            std::vector<double> rows;
            std::vector<int32_t> rowIds;
            for (int32_t h : sel) {   // stage B host part (QSI:238-271): load + decrypt stay on the host
                try {
                    EncryptedPoint ep;
                    if (!index_->loadPointIfActive(index_->idOf(h), &ep)) continue;
                    std::vector<double> v = crypto_->decryptFromPoint(ep, keys_->getVersion(ep.version).key);
                    bool ok = v.size() == q.size();
                    for (double x : v) ok = ok && std::isfinite(x);
                    if (!ok) continue;
                    rows.insert(rows.end(), v.begin(), v.end());
                    rowIds.push_back(h);
                    touched_.insert(index_->idOf(h));
                } catch (...) { continue; }
            }
            lastCandDecrypted_ = static_cast<int>(rowIds.size());
            if (rowIds.empty()) return {};
            const int64_t B = static_cast<int64_t>(rowIds.size());
            std::vector<int32_t> pos(static_cast<size_t>(B)), outIds(static_cast<size_t>(K));
            for (int64_t i = 0; i < B; i++) pos[i] = static_cast<int32_t>(i);
            std::vector<double> outDist(static_cast<size_t>(K));
            int32_t cnt = static_cast<int32_t>(B), outCount = 0, scored = 0;
            check(fspann_refine(index_->nativeContext(), 1, q.data(), rows.data(), FSPANN_F64, B, pos.data(), &cnt, K, outIds.data(),
                                outDist.data(), &outCount, &scored));   // stage B distances + C on the GPU
            std::vector<QueryResult> out;
            lastCandIds_.clear();
            for (int i = 0; i < outCount; i++) {
                out.push_back({index_->idOf(rowIds[outIds[i]]), outDist[i]});
                lastCandIds_.push_back(out.back().id);
            }
            lastReturned_ = outCount;
            if (!retried && (lastReturned_ < K || lastCandDecrypted_ < 10 * K)) {   // QSI:327-337,444-447
                retried = true;
                index_->setProbeOverride(10);
                continue;
            }
            return out;
        }
    }

    int getLastCandTotal() const { return lastCandTotal_; }
    int getLastCandKept() const { return lastCandKept_; }
    int getLastCandDecrypted() const { return lastCandDecrypted_; }
    int getLastReturned() const { return lastReturned_; }
    int getLastUniqueCandidates() const { return lastUnique_; }
    const std::vector<std::string>& getLastFinalResultIds() const { return lastCandIds_; }
    const std::set<std::string>& getTouchedThisSession() const { return touched_; }
    void setRefinementLimit(int limit) { refineOverride_ = limit; }
    void clearRefinementLimit() { refineOverride_ = 0; }
    int getEffectiveRefinementLimit(int def) const { return refineOverride_ > 0 ? refineOverride_ : def; }
    QueryToken deriveToken(const QueryToken* base, int k) {
        if (!tf_) throw IllegalStateException("QueryTokenFactory not available");
        return tf_->derive(base, k);
    }

  private:
    PartitionedIndexService* index_;
    CryptoService* crypto_;
    KeyLifeCycleService* keys_;
    QueryTokenFactory* tf_;
    const SystemConfig* cfg_;
    int lastCandTotal_ = 0, lastCandKept_ = 0, lastCandDecrypted_ = 0, lastReturned_ = 0, lastUnique_ = 0, refineOverride_ = 0;
    std::vector<std::string> lastCandIds_;
    std::set<std::string> touched_;
};

}  // namespace fspann::host
