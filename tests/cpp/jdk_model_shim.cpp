// Test shim (CPU): exposes the product's host-side rare-path code — host/java_hashmap.hpp (literal java.util.HashMap model with
// tree bins) and host/route_replay.hpp (lookupCandidatesWithScores of one query, put by put) — through a plain C ABI so the CPU
// suite can compare it with the oracle's independently written model and with the Python restatement in tests/jdk_hashmap_ref.py.
// Built by tests/test_jdk_hashmap_cpu.py with g++ (also under -fsanitize=address,undefined).
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../fspann-query-system_amd/host/java_hashmap.hpp"
#include "../../fspann-query-system_amd/host/route_replay.hpp"

extern "C" {

// flags: bit 0 = some bin was treeified, bit 1 = unmodelled
int shim_hashmap_order(int32_t initial_capacity, int64_t n, const int32_t* keys, const int32_t* hashes, int decimal, int32_t* out_keys,
                       int32_t* out_cap) {
    fspann::jdk::HashMapModel<fspann::replay::KeyOrderView> m(initial_capacity, fspann::replay::KeyOrderView{decimal != 0});
    for (int64_t i = 0; i < n; i++) m.put(keys[i], hashes[i], i);
    int64_t k = 0;
    m.for_each([&](int32_t key, int64_t) { out_keys[k++] = key; });
    if (out_cap) *out_cap = m.capacity();
    return (m.treeified ? 1 : 0) | (m.unmodelled ? 2 : 0);
}

int shim_compare_decimal(int64_t a, int64_t b) { return fspann::jdk::compare_decimal_strings(a, b); }

struct ShimIndex {
    std::vector<std::vector<int64_t>> mn, mx, off;
    std::vector<std::vector<uint64_t>> rep;
    std::vector<std::vector<int32_t>> ids;
    std::vector<int32_t> jh;
    std::vector<uint32_t> del;
    bool decimal = false, has_del = false;
    int TD = 0, W = 0, S = 64;
};

void* shim_index_create(int TD, int W, int S) {
    ShimIndex* x = new ShimIndex();
    x->TD = TD; x->W = W; x->S = S;
    x->mn.resize(TD); x->mx.resize(TD); x->off.resize(TD); x->rep.resize(TD); x->ids.resize(TD);
    return x;
}
void shim_index_destroy(void* p) { delete static_cast<ShimIndex*>(p); }
void shim_index_set_table(void* p, int td, int64_t np, const int64_t* mn, const int64_t* mx, const uint64_t* rep, const int64_t* off, const int32_t* ids) {
    ShimIndex* x = static_cast<ShimIndex*>(p);
    x->mn[td].assign(mn, mn + np); x->mx[td].assign(mx, mx + np); x->rep[td].assign(rep, rep + np * x->W);
    x->off[td].assign(off, off + np + 1); x->ids[td].assign(ids, ids + off[np]);
}
void shim_index_set_meta(void* p, int64_t n, const int32_t* java_hash, int decimal, const uint8_t* deleted) {
    ShimIndex* x = static_cast<ShimIndex*>(p);
    x->jh.assign(java_hash, java_hash + n);
    x->decimal = decimal != 0;
    x->has_del = deleted != nullptr;
    x->del.assign(static_cast<size_t>((n + 31) / 32), 0u);
    if (deleted) for (int64_t i = 0; i < n; i++) if (deleted[i]) x->del[i >> 5] |= 1u << (i & 31);
}
// returns the list length; ids / score receive min(len, cap) entries; flags: bit 0 treeified, bit 1 unmodelled
int64_t shim_route_query(void* p, const uint64_t* qcodes, int probes, int hard_cap, int64_t cap, int32_t* ids, int32_t* score, int32_t* raw_seen, int* flags) {
    ShimIndex* x = static_cast<ShimIndex*>(p);
    fspann::replay::IndexView v;
    v.TD = x->TD; v.W = x->W; v.S = x->S;
    v.min_key = &x->mn; v.max_key = &x->mx; v.rep = &x->rep; v.id_off = &x->off; v.ids = &x->ids;
    v.java_hash = x->jh.data(); v.decimal_ids = x->decimal; v.deleted_bits = x->has_del ? x->del.data() : nullptr;
    const fspann::replay::Result r = fspann::replay::route_query(v, qcodes, probes, hard_cap);
    const int64_t n = static_cast<int64_t>(r.ids.size());
    for (int64_t i = 0; i < n && i < cap; i++) { ids[i] = r.ids[i]; score[i] = r.score[i]; }
    if (raw_seen) *raw_seen = r.raw_seen;
    if (flags) *flags = (r.treeified ? 1 : 0) | (r.unmodelled ? 2 : 0);
    return n;
}

}  // extern "C"
