// route.hip.h — Route on gfx950: PartitionedIndexService.lookupCandidatesWithScores
// (PIS:592-715) + QueryServiceImpl stage A.5 (QSI:169-214), one workgroup per query.
//
// What the reference does per query (restated): for every table (t,d) in order,
// computeKey(qCode) -> binary search for the partition whose [minKey,maxKey] holds
// the key (idx/GreedyPartitioner.java:101-124) -> best-first expansion to the
// neighbours idx-1/idx+1 ordered by Hamming(q, repCode) through a
// java.util.PriorityQueue (PIS:643-685), at most P partitions -> every id of a
// probed partition gets best[id] = min(best[id], Hamming(q, rep)) in a
// HashMap<String,Long> (PIS:726-753) -> the map's entries, in iteration order,
// stable-sorted by score (PIS:690-696) -> QSI keeps the first B (QSI:208-214).
//
// How it is done here (see DESIGN.md "Route kernel" for the derivations):
//   phase A  one lane per table: key, literal binary search, literal 2-entry heap.
//   phase A2 all lanes: stage the ids of every probed partition into LDS, tuple
//            slot seq = (td*P + step)*S + pos (this IS the reference's insertion
//            order); deleted ids become -1.
//   phase B  tables in order, one barrier per table: open-addressed LDS hash keyed
//            by id holding (min score, first seq).  An id occurs at most once per
//            table, so inside a phase no two lanes touch the same entry; entry
//            creation is the only race and is resolved by atomicCAS on the key.
//            HARD_CAP (PIS:612-615,624,628,657-659) is honoured at probe-step
//            granularity exactly like the reference's loop guards.
//   phase C  Java order key = (score, bucket(id) at the final HashMap capacity,
//            first seq); histogram over score -> cut score s*; entries with
//            score <= s* are compacted and bitonic-sorted in LDS; the first
//            `limit` are written out.
#pragma once
#include "fspann_common.h"

namespace fspann {

struct RouteParams {
    const uint64_t* codes;       // [nq][TD][W]
    const RouteTable* tables;    // [TD]
    const int64_t* keys2;        // [parts][2]
    const uint64_t* rep;         // [parts][W]
    const int32_t* id_off;       // per table nparts+1
    const int32_t* ids;
    const int32_t* java_hash;    // [n_ids]
    const uint32_t* deleted_bits;  // may be null
    int64_t nq;
    int TD, W, P, S;
    int hard_cap, cap0, limit;
    int nbins;                   // bits + 1 score bins
    int ht_size;                 // power of two
    int ht_shift;                // 32 - log2(ht_size)
    int sort_cap_lds;            // LDS sort entries (power of two), 0 => always global
    int max_tuples;              // TD*P*S
    int use_lds_ht;
    uint32_t* g_ht;              // global fallback: per block 2*ht_size u32
    uint64_t* g_sort;            // global fallback: per block g_sort_stride u64
    int64_t g_sort_stride;
    int64_t out_cap;
    int32_t* out_ids;
    int32_t* out_score;
    int32_t* out_count;
    int32_t* out_kept;
    int32_t* out_raw;
};

__device__ __forceinline__ int ham_words(const uint64_t* a, const uint64_t* b, int W) {
    int c = 0;
    for (int i = 0; i < W; i++) c += __popcll(a[i] ^ b[i]);
    return c;
}

// threshold evolution of java.util.HashMap.resize(): returns the table length after
// n insertions into new HashMap<>(cap0-sized) (no treeification, cap0 >= 64).
__device__ __forceinline__ int java_final_cap(int cap0, int n) {
    int cap = cap0;
    int thr = static_cast<int>(static_cast<float>(cap) * 0.75f);
    while (n > thr && cap < (1 << 30)) {
        const int oldCap = cap;
        cap <<= 1;
        thr = (oldCap >= 16) ? (thr << 1) : static_cast<int>(static_cast<float>(cap) * 0.75f);
    }
    return cap;
}

template <typename PtrT>
__device__ __forceinline__ void bitonic_sort_u64(PtrT sb, int n2, int tid, int nthreads) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = sb[i], b = sb[ixj];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) {
                        sb[i] = b;
                        sb[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <bool kLdsHT>
__global__ __launch_bounds__(512) void route_kernel(RouteParams prm) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int TD = prm.TD, W = prm.W, P = prm.P, S = prm.S;

    // ---- carve LDS -----------------------------------------------------------
    uint64_t* lds_sort = reinterpret_cast<uint64_t*>(smem);
    size_t o = static_cast<size_t>(prm.sort_cap_lds) * 8;
    uint32_t* lds_htk = reinterpret_cast<uint32_t*>(smem + o);
    if (kLdsHT) o += static_cast<size_t>(prm.ht_size) * 8;
    uint32_t* lds_htv = lds_htk + prm.ht_size;
    int32_t* tup = reinterpret_cast<int32_t*>(smem + o);
    o += static_cast<size_t>(prm.max_tuples) * 4;
    int32_t* hist = reinterpret_cast<int32_t*>(smem + o);
    o += static_cast<size_t>(prm.nbins) * 4;
    int32_t* probe_part = reinterpret_cast<int32_t*>(smem + o);
    o += static_cast<size_t>(TD) * P * 4;
    int32_t* probe_dist = reinterpret_cast<int32_t*>(smem + o);
    o += static_cast<size_t>(TD) * P * 4;
    int32_t* nprobe = reinterpret_cast<int32_t*>(smem + o);

    __shared__ int s_phase[3];
    __shared__ int s_raw;
    __shared__ int s_fill;
    __shared__ int s_star;
    __shared__ int s_sel;

    uint32_t* htk = kLdsHT ? lds_htk : prm.g_ht + static_cast<size_t>(blockIdx.x) * 2 * prm.ht_size;
    uint32_t* htv = kLdsHT ? lds_htv : htk + prm.ht_size;
    const uint32_t ht_mask = static_cast<uint32_t>(prm.ht_size - 1);

    for (int64_t qi = blockIdx.x; qi < prm.nq; qi += gridDim.x) {
        // ---- reset ------------------------------------------------------------
        for (int i = tid; i < prm.ht_size; i += nthreads) htk[i] = kEmptyKey;
        for (int i = tid; i < prm.nbins; i += nthreads) hist[i] = 0;
        if (tid < 3) s_phase[tid] = 0;
        if (tid == 0) { s_raw = 0; s_fill = 0; }

        // ---- phase A: per-table search + probe order ---------------------------
        for (int td = tid; td < TD; td += nthreads) {
            const RouteTable tb = prm.tables[td];
            const uint64_t* qc = prm.codes + (qi * TD + td) * W;
            int np = 0;
            if (tb.nparts > 0) {
                // GreedyPartitioner.computeKey: bit i of the code -> key bit 62-i, i < 63
                const int64_t qKey = static_cast<int64_t>(__brevll(qc[0]) >> 1);
                const int64_t* k2 = prm.keys2 + tb.part_base * 2;
                int lo = 0, hi = tb.nparts - 1, center = -1;
                while (lo <= hi) {
                    const int mid = static_cast<int>((static_cast<unsigned>(lo) + static_cast<unsigned>(hi)) >> 1);
                    const longlong2 mm = *reinterpret_cast<const longlong2*>(k2 + 2 * static_cast<int64_t>(mid));
                    if (qKey < mm.x) hi = mid - 1;
                    else if (qKey > mm.y) lo = mid + 1;
                    else { center = mid; break; }
                }
                if (center < 0) {
                    if (lo <= 0) center = 0;
                    else if (lo >= tb.nparts) center = tb.nparts - 1;
                    else {
                        const longlong2 L = *reinterpret_cast<const longlong2*>(k2 + 2 * static_cast<int64_t>(lo - 1));
                        const longlong2 R = *reinterpret_cast<const longlong2*>(k2 + 2 * static_cast<int64_t>(lo));
                        const int64_t dl = (qKey < L.x) ? (L.x - qKey) : ((qKey > L.y) ? (qKey - L.y) : 0);
                        const int64_t dr = (qKey < R.x) ? (R.x - qKey) : ((qKey > R.y) ? (qKey - R.y) : 0);
                        center = (dl <= dr) ? (lo - 1) : lo;
                    }
                }
                // java.util.PriorityQueue with <= 2 live entries (left / right frontier).
                // offer(): the newcomer becomes the root only if STRICTLY smaller (siftUp);
                // poll(): the survivor becomes the root (siftDown on one element).
                const uint64_t* repb = prm.rep + tb.part_base * W;
                int h_idx0 = center, h_d0 = ham_words(qc, repb + static_cast<int64_t>(center) * W, W);
                int h_idx1 = 0, h_d1 = 0, hn = 1;
                int vlo = center, vhi = center;
                while (hn > 0 && np < P) {
                    const int cur = h_idx0, curd = h_d0;
                    hn--;
                    if (hn == 1) { h_idx0 = h_idx1; h_d0 = h_d1; }
                    probe_part[td * P + np] = cur;
                    probe_dist[td * P + np] = curd;
                    np++;
                    const int left = cur - 1;
                    if (left >= 0 && left < vlo) {
                        vlo = left;
                        const int dd = ham_words(qc, repb + static_cast<int64_t>(left) * W, W);
                        if (hn == 0) { h_idx0 = left; h_d0 = dd; }
                        else if (dd < h_d0) { h_idx1 = h_idx0; h_d1 = h_d0; h_idx0 = left; h_d0 = dd; }
                        else { h_idx1 = left; h_d1 = dd; }
                        hn++;
                    }
                    const int right = cur + 1;
                    if (right < tb.nparts && right > vhi) {
                        vhi = right;
                        const int dd = ham_words(qc, repb + static_cast<int64_t>(right) * W, W);
                        if (hn == 0) { h_idx0 = right; h_d0 = dd; }
                        else if (dd < h_d0) { h_idx1 = h_idx0; h_d1 = h_d0; h_idx0 = right; h_d0 = dd; }
                        else { h_idx1 = right; h_d1 = dd; }
                        hn++;
                    }
                }
            }
            nprobe[td] = np;
        }
        __syncthreads();

        // ---- phase A2: stage ids of all probed partitions ----------------------
        for (int j = tid; j < prm.max_tuples; j += nthreads) {
            const int ts = j / S, pos = j - ts * S;
            const int td = ts / P, step = ts - td * P;
            int32_t id = -1;
            if (step < nprobe[td]) {
                const RouteTable tb = prm.tables[td];
                const int part = probe_part[ts];
                const int32_t* off = prm.id_off + tb.off_base + part;
                const int b0 = off[0], b1 = off[1];
                if (pos < b1 - b0) {
                    id = prm.ids[tb.ids_base + b0 + pos];
                    if (prm.deleted_bits && ((prm.deleted_bits[id >> 5] >> (id & 31)) & 1u)) id = -1;
                }
            }
            tup[j] = id;
        }
        __syncthreads();

        // ---- phase B: ordered insertion -----------------------------------------
        int size = 0;   // bestScore.size(), identical in every lane
        int phase = 0;  // rotation index into s_phase
        bool stop = false;
        for (int td = 0; td < TD && !stop; td++) {
            const int np = nprobe[td];
            if (np == 0) continue;
            if (size >= prm.hard_cap) break;  // PIS:624,628
            const bool whole = (size + (np - 1) * S < prm.hard_cap);
            const int nsub = whole ? 1 : np;
            for (int sub = 0; sub < nsub; sub++) {
                if (size >= prm.hard_cap) { stop = true; break; }  // PIS:657-659
                const int j0 = (td * P + (whole ? 0 : sub)) * S;
                const int j1 = whole ? (td * P + np) * S : j0 + S;
                int created = 0, touched = 0;
                for (int j = j0 + tid; j < j1; j += nthreads) {
                    const int32_t id = tup[j];
                    if (id < 0) continue;
                    const uint32_t score = static_cast<uint32_t>(probe_dist[j / S]);
                    uint32_t slot = (static_cast<uint32_t>(id) * 2654435761u) >> prm.ht_shift;
                    while (true) {
                        uint32_t kk = htk[slot];
                        if (kk == kEmptyKey) {
                            kk = atomicCAS(&htk[slot], kEmptyKey, static_cast<uint32_t>(id));
                            if (kk == kEmptyKey) {  // created: bestScore.put(id, score) of a new key
                                htv[slot] = (score << kSeqBits) | static_cast<uint32_t>(j);
                                created++;
                                break;
                            }
                        }
                        if (kk == static_cast<uint32_t>(id)) {  // prev != null
                            const uint32_t cur = htv[slot];
                            if (score < (cur >> kSeqBits)) {      // score < prev -> put, position kept
                                htv[slot] = (score << kSeqBits) | (cur & kSeqMask);
                                touched++;
                            }
                            break;
                        }
                        slot = (slot + 1) & ht_mask;
                    }
                }
                if (created) atomicAdd(&s_phase[phase % 3], created);
                if (created + touched) atomicAdd(&s_raw, created + touched);
                if (tid == 0) s_phase[(phase + 1) % 3] = 0;
                __syncthreads();
                size += s_phase[phase % 3];
                phase++;
            }
        }
        __syncthreads();

        // ---- phase C: order + select ----------------------------------------------
        const int n = size;
        const int capf = java_final_cap(prm.cap0, n);
        const uint32_t bmask = static_cast<uint32_t>(capf - 1);
        for (int i = tid; i < prm.ht_size; i += nthreads)
            if (htk[i] != kEmptyKey) atomicAdd(&hist[htv[i] >> kSeqBits], 1);
        __syncthreads();
        if (tid == 0) {
            int cum = 0, star = prm.nbins - 1;
            if (n > prm.limit) {
                for (int b = 0; b < prm.nbins; b++) {
                    cum += hist[b];
                    if (cum >= prm.limit) { star = b; break; }
                }
            } else {
                cum = n;
            }
            s_star = star;
            s_sel = cum;  // entries with score <= s*
        }
        __syncthreads();
        const int star = s_star, nsel = s_sel;
        int n2 = 1;
        while (n2 < nsel) n2 <<= 1;
        const bool lds_sort_ok = (n2 <= prm.sort_cap_lds);
        uint64_t* gs = prm.g_sort + static_cast<int64_t>(blockIdx.x) * prm.g_sort_stride;
        for (int i = tid; i < prm.ht_size; i += nthreads) {
            const uint32_t id = htk[i];
            if (id == kEmptyKey) continue;
            const uint32_t v = htv[i];
            const uint32_t sc = v >> kSeqBits;
            if (static_cast<int>(sc) > star) continue;
            uint32_t h = static_cast<uint32_t>(prm.java_hash[id]);
            h ^= (h >> 16);  // HashMap.hash(): spread
            const uint64_t key = (static_cast<uint64_t>(sc) << (kBucketBits + kSeqBits)) |
                                 (static_cast<uint64_t>(h & bmask) << kSeqBits) | (v & kSeqMask);
            const int pos = atomicAdd(&s_fill, 1);
            if (lds_sort_ok) lds_sort[pos] = key; else gs[pos] = key;
        }
        for (int i = nsel + tid; i < n2; i += nthreads) {
            if (lds_sort_ok) lds_sort[i] = ~0ull; else gs[i] = ~0ull;
        }
        __syncthreads();
        if (lds_sort_ok) bitonic_sort_u64(lds_sort, n2, tid, nthreads);
        else bitonic_sort_u64(gs, n2, tid, nthreads);

        const int nout = min(nsel, prm.limit);
        for (int i = tid; i < nout; i += nthreads) {
            const uint64_t key = lds_sort_ok ? lds_sort[i] : gs[i];
            const int32_t id = tup[static_cast<uint32_t>(key) & kSeqMask];
            prm.out_ids[qi * prm.out_cap + i] = id;
            if (prm.out_score) prm.out_score[qi * prm.out_cap + i] = static_cast<int32_t>(key >> (kBucketBits + kSeqBits));
        }
        if (tid == 0) {
            prm.out_count[qi] = nout;
            if (prm.out_kept) prm.out_kept[qi] = n;
            if (prm.out_raw) prm.out_raw[qi] = s_raw;
        }
        __syncthreads();
    }
}

}  // namespace fspann
