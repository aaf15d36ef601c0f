// route_replay.hpp — PartitionedIndexService.lookupCandidatesWithScores (PIS:592-715) for ONE query, put by put, on the host.
// Product code (pure host C++17, no HIP).  This is the library's RARE path: the Route kernels derive the iteration order of
// HashMap<String,Long> bestScore in closed form, which is exact while every bin of the map is a plain chain; a query whose map
// would treeify a bin (detected exactly by the full select, count = -1) is finished here with the literal JDK model of
// java_hashmap.hpp — traversal, HARD_CAP rule and counters exactly as the reference runs them — instead of being refused.
// ~0.3 % of the queries at the reference's shipped profiles (20-36 k ids in 32 768 / 65 536 bins), ~2e-9 at BASELINE config #2.
// Never includes or links anything under oracle/.
#pragma once
#include <algorithm>
#include <cstdint>
#include <queue>
#include <vector>

#include "java_hashmap.hpp"

namespace fspann {
namespace replay {

// Read-only view of a frozen index as the context's host mirror holds it (one entry per (t,d) table).
struct IndexView {
    int TD = 0, W = 0, S = 64;
    const std::vector<std::vector<int64_t>>* min_key = nullptr;
    const std::vector<std::vector<int64_t>>* max_key = nullptr;
    const std::vector<std::vector<uint64_t>>* rep = nullptr;       // [nparts][W]
    const std::vector<std::vector<int64_t>>* id_off = nullptr;     // [nparts + 1]
    const std::vector<std::vector<int32_t>>* ids = nullptr;
    const int32_t* java_hash = nullptr;                            // String.hashCode per handle
    bool decimal_ids = false;                                      // ids are Long.toString(handle): compareTo is computable
    const uint32_t* deleted_bits = nullptr;                        // metadata.isDeleted mirror (may be null)
};

struct KeyOrderView {
    bool decimal;
    int operator()(int32_t a, int32_t b) const { return decimal ? jdk::compare_decimal_strings(a, b) : 0; }
};

struct Result {
    std::vector<int32_t> ids, score;   // the whole list: HashMap iteration order, stable-sorted by score (PIS:690-696)
    int32_t raw_seen = 0;              // PIS.getLastRawCandidateCount
    bool unmodelled = false;           // a tree bin had to order equal hashCodes of ids whose Strings the library does not know
    bool treeified = false;
};

// java.util.PriorityQueue<long[]>(comparingLong(a -> a[1])): array heap, strict comparisons (PIS:643-644)
struct ProbeHeap {
    struct E { int64_t idx, dist; };
    std::vector<E> q;
    void add(E x) {
        size_t k = q.size();
        q.push_back(x);
        while (k > 0) {
            const size_t parent = (k - 1) >> 1;
            if (!(x.dist < q[parent].dist)) break;
            q[k] = q[parent];
            k = parent;
        }
        q[k] = x;
    }
    E poll() {
        const E result = q[0];
        const size_t n = q.size() - 1;
        const E x = q[n];
        q.pop_back();
        if (n > 0) {
            size_t k = 0;
            const size_t half = n >> 1;
            while (k < half) {
                size_t child = 2 * k + 1;
                const size_t right = child + 1;
                if (right < n && q[child].dist > q[right].dist) child = right;
                if (x.dist <= q[child].dist) break;
                q[k] = q[child];
                k = child;
            }
            q[k] = x;
        }
        return result;
    }
};

inline int64_t compute_key(const uint64_t* w) {            // GreedyPartitioner.computeKey: code bit i -> key bit 62 - i, i < 63
    uint64_t x = w[0], rev = 0;
    for (int b = 0; b < 64; b++) { rev = (rev << 1) | (x & 1); x >>= 1; }
    return static_cast<int64_t>(rev >> 1);
}
inline int64_t hamming(const uint64_t* a, const uint64_t* b, int W) {
    int64_t c = 0;
    for (int i = 0; i < W; i++) c += __builtin_popcountll(a[i] ^ b[i]);
    return c;
}
inline int find_nearest_partition(const std::vector<int64_t>& mn, const std::vector<int64_t>& mx, int64_t qKey) {   // GreedyPartitioner.java:101-124
    const int n = static_cast<int>(mn.size());
    if (n == 0) return 0;
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = static_cast<int>((static_cast<unsigned>(lo) + static_cast<unsigned>(hi)) >> 1);
        if (qKey < mn[mid]) hi = mid - 1;
        else if (qKey > mx[mid]) lo = mid + 1;
        else return mid;
    }
    if (lo <= 0) return 0;
    if (lo >= n) return n - 1;
    auto dist = [&](int p) -> int64_t { return qKey < mn[p] ? mn[p] - qKey : (qKey > mx[p] ? qKey - mx[p] : 0); };
    return dist(lo - 1) <= dist(lo) ? lo - 1 : lo;
}

// qcodes = [TD][W].  probes = effectiveMaxProbes(); hard_cap = max(maxGlobalCandidates, refinementLimit) (PIS:612-615).
inline Result route_query(const IndexView& v, const uint64_t* qcodes, int probes, int hard_cap) {
    Result out;
    jdk::HashMapModel<KeyOrderView> best(std::min(hard_cap, 1 << 16), KeyOrderView{v.decimal_ids});     // PIS:619
    best.reserve(static_cast<size_t>(std::min<int64_t>(static_cast<int64_t>(v.TD) * probes * v.S, hard_cap + v.S)));
    auto deleted = [&](int32_t id) { return v.deleted_bits && ((v.deleted_bits[id >> 5] >> (id & 31)) & 1u); };
    std::vector<char> visited;
    for (int td = 0; td < v.TD && best.size() < hard_cap; td++) {
        const auto& mn = (*v.min_key)[td];
        const auto& mx = (*v.max_key)[td];
        const int nparts = static_cast<int>(mn.size());
        if (nparts == 0) continue;
        const uint64_t* q = qcodes + static_cast<size_t>(td) * v.W;
        const uint64_t* rep = (*v.rep)[td].data();
        const int center = find_nearest_partition(mn, mx, compute_key(q));
        ProbeHeap pq;
        visited.assign(static_cast<size_t>(nparts), 0);
        pq.add({center, hamming(q, rep + static_cast<size_t>(center) * v.W, v.W)});
        visited[center] = 1;
        int used = 0;
        while (!pq.q.empty() && used < probes && best.size() < hard_cap) {
            const ProbeHeap::E cur = pq.poll();
            const int idx = static_cast<int>(cur.idx);
            used++;
            {   // collectPartitionOrdered (PIS:726-753)
                const int64_t part_dist = hamming(q, rep + static_cast<size_t>(idx) * v.W, v.W);
                const int64_t b0 = (*v.id_off)[td][idx], b1 = (*v.id_off)[td][idx + 1];
                const int32_t* ids = (*v.ids)[td].data();
                for (int64_t i = b0; i < b1; i++) {
                    const int32_t id = ids[i];
                    if (deleted(id)) continue;
                    const int32_t jh = v.java_hash[id];
                    const int64_t* prev = best.get(id, jh);
                    if (prev == nullptr || part_dist < *prev) {
                        best.put(id, jh, part_dist);
                        out.raw_seen++;
                    }
                }
            }
            const int left = idx - 1;
            if (left >= 0 && !visited[left]) { visited[left] = 1; pq.add({left, hamming(q, rep + static_cast<size_t>(left) * v.W, v.W)}); }
            const int right = idx + 1;
            if (right < nparts && !visited[right]) { visited[right] = 1; pq.add({right, hamming(q, rep + static_cast<size_t>(right) * v.W, v.W)}); }
        }
    }
    struct Ent { int32_t id; int64_t score; };
    std::vector<Ent> list;
    list.reserve(static_cast<size_t>(best.size()));
    best.for_each([&](int32_t key, int64_t val) { list.push_back({key, val}); });
    std::stable_sort(list.begin(), list.end(), [](const Ent& a, const Ent& b) { return a.score < b.score; });   // List.sort is stable
    out.ids.resize(list.size());
    out.score.resize(list.size());
    for (size_t i = 0; i < list.size(); i++) { out.ids[i] = list[i].id; out.score[i] = static_cast<int32_t>(list[i].score); }
    out.unmodelled = best.unmodelled;
    out.treeified = best.treeified;
    return out;
}

}  // namespace replay
}  // namespace fspann
