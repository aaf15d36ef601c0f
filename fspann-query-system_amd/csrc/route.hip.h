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
// How it is done here (derivations in DESIGN.md "Route kernel"); A is its own kernel
// (route_probe_kernel, one wave per (query, table) -> full occupancy for the dependent
// L2 rounds), A2..C are route_select_kernel (one workgroup per query, all state in LDS):
//   A   one lane GROUP (G = 64 / tables-per-wave lanes) per table: G-ary search for the run [a,b] of partitions whose
//       key range holds the query key, then the reference's binary search is
//       replayed arithmetically against (a,b) — same `mid` sequence, no dependent
//       loads; Hamming distances of the 2P-1 reachable partitions are fetched in
//       one round and the 2-entry PriorityQueue is replayed from LDS.
//   A2  all lanes: ids of every probed partition -> LDS tuple slots,
//       seq = (td*P + step)*S + pos (= the reference's insertion order).
//   B1  (fused with A2) all tuples in parallel: open-addressed LDS hash, 4-byte
//       entries holding the FIRST seq of an id (atomicMin); identity is checked
//       through tup[seq]; double hashing.  Every atomicMin that meets an occupied
//       entry exposes exactly one repeated occurrence (the larger seq): those go to
//       a small list; the score histogram of first occurrences is kept up to date.
//   B2  HARD_CAP (PIS:612-615,624,628,657-659; rare): distinct-count per probe step
//       -> first step at which size >= cap; later steps never ran.
//   B3  repeated occurrences: min score and the number of strict improvements in
//       table order (= rawSeen - created, PIS:744-750), all repeats in parallel.
//   C   Java order key (score | HashMap bucket at the final capacity | first seq):
//       radix-select on score, then on the bucket's top bits, down to <= ~limit
//       candidates; all-pairs rank sort (no barriers) writes the first `limit`.
#pragma once
#include "fspann_common.h"

namespace fspann {

struct RouteParams {
    const uint64_t* codes;         // [nq][TD][W]
    const RouteTable* tables;      // [TD]
    const int64_t* recs;           // [parts][rec_words] partition records {minKey, maxKey, rep[W], id offset | size << 32}
    int rec_words;
    const int2* dir;               // radix directory: per table 2^dir_bits + 1 pairs {first partition with maxKey >= p << s,
    int dir_bits;                  //   first partition with minKey >= p << s}, s = 63 - dir_bits; null: search from scratch
    const int32_t* ids;
    const int32_t* java_hash;      // [n_ids]
    const uint32_t* deleted_bits;  // may be null
    int64_t nq;
    int TD, W, P, S, S_shift;      // S_shift = log2(S) or -1
    int hard_cap, cap0, limit;
    int need_cap;                  // 1 if TD*P*S >= hard_cap (the cap can trigger)
    int nbins;                     // bits + 1 score bins (<= 1024)
    int ht_size, ht_shift;         // power of two; shift = 32 - log2
    int seq_bits;                  // ceil(log2(max_tuples)): hash entries are (tag << seq_bits) | seq
    int sort_cap;                  // entries of the per-block sort buffer (LDS or global)
    int max_tuples;                // TD*P*S
    unsigned char* g_scratch;      // global fallback arena (per block g_stride bytes) when !kLds
    int64_t g_stride;
    uint64_t* g_sort;              // global sort fallback (per block g_sort_stride u64), may be null
    int64_t g_sort_stride;
    uint32_t* g_sub;               // long lists: (bucket | seq) sub-keys grouped by score (per block g_sub_stride u32), may be null
    int64_t g_sub_stride;
    int lds_sort_words;            // !kLds: u32 words of LDS behind the small arrays for sorting one score group (kLds: the hash table's)
    int dev_flags;                 // FSPANN_ROUTE_DEVFLAGS (dev A/B): 2 = no nibble-counter treeify check
    int slice_bits, slice_ht;      // !kLds: the hash is built in 2^slice_bits passes over slices of the id space, each in an LDS table of
                                   //   slice_ht slots (a power of two <= lds_sort_words); slice_ht == 0: one table in the arena / LDS
    int wave_sort;                 // 1: groups are sorted by single waves on their own LDS slices (0: by the whole workgroup, one by one)
    int64_t out_cap;
    int32_t* out_ids;
    int32_t* out_score;
    int32_t* out_count;
    int32_t* out_kept;
    int32_t* out_raw;
    int decimal_ids;               // 1: ids are Long.toString(handle) -> hash computed arithmetically
    // bounded ("lazy") select, route_lazy.hip.h
    const int32_t* inv;            // [TD][n_ids] position of an id in table td's id list (-1 = absent)
    const uint64_t* ids_bk;        // ids of every partition as (id << 32 | bucket field), bucket-sorted within the partition
    int64_t n_ids;
    const uint16_t* bin16;         // [parts][1 << bin16_shift] HashMap bin of every id, partition order (null: no exact treeify check)
    int bin16_shift;
    int lazy_cap;                  // tuples one query may insert before it is handed to route_select_kernel
    int lz_ht_size, lz_ht_shift;
    int32_t* ovf_next;             // the OTHER overflow counter: zeroed by the lazy kernel for the next call (ping-pong)
    int4* probe_g;                 // [nq][TD*P] / [nq][TD] global probe lists: written by route_probe_kernel, or by the bounded
    int32_t* nprobe_g;             //   select itself (fused probe) when it hands a query to route_select_query
    int probe_G;                   // fused probe: lanes per table (0: the probe list comes from route_probe_kernel)
    int32_t* ovf_count;            // overflow list written by the bounded select ...
    int32_t* ovf_list;
    const int32_t* qcount;         // ... and consumed by route_select_kernel (list mode: only these queries)
    const int32_t* qlist;
    int32_t* unmodelled;           // context-wide count of queries whose HashMap would have treeified a bin (out_count = -1)
    long long* dbg;                // FSPANN_DEBUG_STAMPS builds: [grid][16] wall_clock64 stamps of each block's first query (else unused)
};

__device__ __forceinline__ int ham_words(const uint64_t* a, const uint64_t* b, int W) {
    int c = 0;
    for (int i = 0; i < W; i++) c += __popcll(a[i] ^ b[i]);
    return c;
}

// java.util.HashMap.resize() threshold evolution: table length after n insertions into
// new HashMap<>(cap0-sized) (no treeification, cap0 >= 64).
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

__device__ __forceinline__ void bitonic_sort_u32(uint32_t* sb, int n2, int tid, int nthreads) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t a = sb[i], b = sb[ixj];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) { sb[i] = b; sb[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// the same network run by ONE wave on its own LDS slice: no workgroup barrier anywhere (LDS operations of a wave complete in order)
__device__ __forceinline__ void bitonic_sort_u32_wave(uint32_t* sb, int n2, int lane) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < n2; i += 64) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t a = sb[i], b = sb[ixj];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) { sb[i] = b; sb[ixj] = a; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

constexpr uint32_t kHtEmpty = 0xFFFFFFFFu;
constexpr int32_t kRouteUnmodelled = -1;   // out_count: a HashMap bin would be treeified, the JVM's order is not modelled
constexpr int32_t kRoutePending = -2;      // out_count: handed over by the bounded select, the full select has not run yet
constexpr uint16_t kFirstFlag = 0x8000u;  // tscore: first occurrence of its id (score in the low 14 bits)
constexpr int kRankSortMax = 1024;

// String.hashCode(Long.toString(v)), v >= 0: h = sum (48 + digit_i) * 31^i over digits from the LSB.
__device__ __forceinline__ uint32_t decimal_string_hash_dev(uint32_t v) {
    uint32_t h = 0, pw = 1;
    do {
        const uint32_t q = v / 10u, dgt = v - q * 10u;
        h += (48u + dgt) * pw;
        pw *= 31u;
        v = q;
    } while (v);
    return h;
}

constexpr int kProbeThreads = 256;
constexpr int kDupListMax = 768;
constexpr int kStageU = 12;                 // id loads in flight per lane while staging (one round trip for 10 tuples per lane)
constexpr uint16_t kLiveFlag = 0x4000u;   // tscore: tuple slot holds a live (non-deleted) id

// First index b in [0, nb) with prefix(b) >= need, computed by ONE wave (nb <= 1024).
// Returns b (nb - 1 if the total is smaller) and the count strictly before b through *before.
__device__ __forceinline__ int wave_find_cut(const int32_t* bins, int nb, int need, int lane, int* before) {
    const int per = (nb + 63) >> 6;
    const int b0 = lane * per;
    int sum = 0;
    for (int i = 0; i < per; i++) sum += (b0 + i < nb) ? bins[b0 + i] : 0;
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    const int excl = incl - sum;
    const bool mine = (incl >= need) && (excl < need);
    const unsigned long long bm = __ballot(mine);
    int res_b = nb - 1, res_before = 0;
    if (bm == 0) {  // total < need: everything before the last bin
        res_before = __shfl(incl, 63) - bins[nb - 1];
    } else {
        const int src = __ffsll(static_cast<long long>(bm)) - 1;
        int cum = excl, bb = nb - 1;
        if (lane == src) {
            for (int i = 0; i < per && b0 + i < nb; i++) {
                if (cum + bins[b0 + i] >= need) { bb = b0 + i; break; }
                cum += bins[b0 + i];
            }
        }
        res_b = __shfl(bb, src);
        res_before = __shfl(cum, src);
    }
    *before = res_before;
    return res_b;
}

// The best-first expansion of PIS:643-685 for one table, replayed from the group's LDS scratch w3 = [2P-1][3]
// {Hamming(q, rep), id offset, size} of the partitions center-(P-1) .. center+(P-1).  Every lane of the group runs it (the
// trip count is group-uniform); lane 0 writes the list.  Returns the number of partitions probed.
template <typename OutPtr>
__device__ __forceinline__ int route_probe_replay(const RouteTable& tb, const int P, const int nd, const int center, const int gl,
                                                  const int32_t* w3, OutPtr po) {
    // java.util.PriorityQueue with <= 2 live entries: offer() makes the newcomer the root only
    // if STRICTLY smaller (siftUp); poll() promotes the survivor (siftDown on one element).
    int np = 0;
    int h_idx0 = center, h_d0 = w3[(P - 1) * 3];
    int h_idx1 = 0, h_d1 = 0, hn = 1;
    int vlo = center, vhi = center;
    while (hn > 0 && np < P) {
        const int cur = h_idx0, curd = h_d0;
        hn--;
        if (hn == 1) { h_idx0 = h_idx1; h_d0 = h_d1; }
        if (gl == 0) {
            const int ci = cur - center + (P - 1);
            po[np] = make_int4(cur, curd, w3[ci * 3 + 1], w3[ci * 3 + 2]);
        }
        np++;
        const int left = cur - 1;
        if (left >= 0 && left < vlo) {
            vlo = left;
            const int li = left - center + (P - 1);
            const int dd = (li >= 0 && li < nd) ? w3[li * 3] : 0;  // out of reach: never polled
            if (hn == 0) { h_idx0 = left; h_d0 = dd; }
            else if (dd < h_d0) { h_idx1 = h_idx0; h_d1 = h_d0; h_idx0 = left; h_d0 = dd; }
            else { h_idx1 = left; h_d1 = dd; }
            hn++;
        }
        const int right = cur + 1;
        if (right < tb.nparts && right > vhi) {
            vhi = right;
            const int ri = right - center + (P - 1);
            const int dd = (ri >= 0 && ri < nd) ? w3[ri * 3] : 0;
            if (hn == 0) { h_idx0 = right; h_d0 = dd; }
            else if (dd < h_d0) { h_idx1 = h_idx0; h_d1 = h_d0; h_idx0 = right; h_d0 = dd; }
            else { h_idx1 = right; h_d1 = dd; }
            hn++;
        }
    }
    return np;
}

// ------------------------------------------------------------------------------------------
// Kernel 1: search + probe order.  One lane group (G lanes) per (query, table).
// probe_out[(q*TD + td)*P + step] = {partition, Hamming, id_off b0, size}; nprobe_out[q*TD + td].
// ------------------------------------------------------------------------------------------
// Probe list of ONE (query, table) by one group of G lanes (G | 64, group-aligned inside a wave): the partitions
// PIS.lookupCandidatesWithScores visits for this table, in the reference's order.  `act` = the group has a table to work on
// (every lane of the wave must call: ballots inside).  w3 = LDS scratch of the group, (2P-1)*3 ints.  The list goes to
// po[0..np) (LDS or global); returns np.
// kP / kW (0 = run-time values): probes per table and code words as constants (the shape-specialised bounded select, route_lazy.hip.h).
template <typename OutPtr, int kP = 0, int kW = 0>
__device__ __forceinline__ int route_probe_table(const RouteParams& prm, bool act, const uint64_t* qc, const RouteTable tb, int G, int gl,
                                                 int grp_in_wave, int32_t* w3, OutPtr po, long long* pstamp = nullptr) {
#ifdef FSPANN_DEBUG_STAMPS
    int pst_i = 0;
#define PROBE_STAMP() do { if (pstamp && pst_i < 12) pstamp[pst_i++] = wall_clock64(); } while (0)
#else
#define PROBE_STAMP() do { (void)pstamp; } while (0)
#endif
    PROBE_STAMP();
    const int W = kW > 0 ? kW : prm.W, P = kP > 0 ? kP : prm.P;
    const int nd = 2 * P - 1;
    act = act && tb.nparts > 0;
    // G is a power of two (both callers): shifts instead of divisions by a run-time value — a 32-bit division is ~25 vector
    // instructions, and the two segment lengths of every search round were a sixth of the bounded select's probe phase
    const int lgG = 31 - __clz(G);
    const int gshift = grp_in_wave << lgG;
    const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << gshift);
    // GreedyPartitioner.computeKey: code bit i -> key bit 62-i (i < 63)
    const int64_t qKey = act ? static_cast<int64_t>(__brevll(qc[0]) >> 1) : 0;
#ifdef FSPANN_DEBUG_STAMPS
    if (pstamp) { asm volatile("" :: "v"(static_cast<int>(qKey)), "v"(tb.nparts)); PROBE_STAMP(); }   // code word and table record are here
#endif
    const int RW = kW > 0 ? ((3 + kW + 1) & ~1) : prm.rec_words;     // (= the host's rec_words for W code words)
    const int64_t* recs = prm.recs + tb.part_base * RW;     // this table's partition records
    // a = first partition with maxKey >= qKey ; e = first partition with minKey > qKey.
    // Invariant of both searches: answer in [lo, hi], hi == nparts or pred(hi) true.
    int loA = 0, hiA = act ? tb.nparts : 0, loE = 0, hiE = hiA;
    if (prm.dir && act) {
        // radix directory on the key's top bits: ONE dependent load instead of the first rounds of the search.  Keys are
        // monotone over the partitions (checked when the directory is built), so for a query key with prefix p
        // a lies in [A[p], A[p+1]] and e in [E[p], E[p+1]], and the predicate holds at the upper ends (or they are nparts).
        const int2* dd = prm.dir + tb.dir_base + (qKey >> (63 - prm.dir_bits));
        const int2 d0 = dd[0], d1 = dd[1];
        loA = d0.x; hiA = d1.x; loE = d0.y; hiE = d1.y;
    }
    asm volatile("" :: "v"(loA), "v"(hiA));
    PROBE_STAMP();     // 1: directory entry is here
    // G-ary search, both at once, from the directory's (or the whole table's) bounds.  (A variant that fetched key ranges, codes
    // and id ranges of a whole window of <= 32 partitions in one round and resolved everything from registers was measured and
    // dropped: the brackets are rarely that narrow — LSH keys are skewed — and its 14 extra registers per lane were the
    // register peak of the bounded select.)
    while (__any((hiA > loA) || (hiE > loE))) {
        const int stA = (hiA - loA + G - 1) >> lgG, stE = (hiE - loE + G - 1) >> lgG;
        const int sA = loA + gl * stA, sE = loE + gl * stE;  // my segment starts
        bool pA = false, pE = false;
        // (32-bit record offsets: a table holds < 2^31 / rec_words partitions)
        if (hiA > loA && sA < hiA) pA = recs[(min(sA + stA, hiA) - 1) * RW + 1] >= qKey;
        if (hiE > loE && sE < hiE) pE = recs[(min(sE + stE, hiE) - 1) * RW] > qKey;
        const unsigned long long bA = __ballot(pA) & gmask, bE = __ballot(pE) & gmask;
        if (hiA > loA) {
            if (bA == 0) loA = hiA;
            else {
                const int f = (__ffsll(static_cast<long long>(bA)) - 1) - gshift;
                const int nlo = loA + f * stA;
                hiA = min(nlo + stA, hiA) - 1;
                loA = nlo;
            }
        }
        if (hiE > loE) {
            if (bE == 0) loE = hiE;
            else {
                const int f = (__ffsll(static_cast<long long>(bE)) - 1) - gshift;
                const int nlo = loE + f * stE;
                hiE = min(nlo + stE, hiE) - 1;
                loE = nlo;
            }
        }
        PROBE_STAMP();     // 2..: one per search round
    }
    int center = 0;
    if (act) {
        const int a = loA, b = loE - 1;
        if (a <= b) {
            // replay findNearestPartition's loop: mid > b <=> qKey < minKey[mid]; mid < a <=> qKey > maxKey[mid]
            int lo = 0, hi = tb.nparts - 1;
            center = a;
            while (lo <= hi) {
                const int mid = static_cast<int>((static_cast<unsigned>(lo) + static_cast<unsigned>(hi)) >> 1);
                if (mid > b) hi = mid - 1;
                else if (mid < a) lo = mid + 1;
                else { center = mid; break; }
            }
        } else {
            const int lo = a;  // the loop ends with lo = first partition with minKey > qKey
            if (lo <= 0) center = 0;
            else if (lo >= tb.nparts) center = tb.nparts - 1;
            else {
                const int64_t lmax = recs[(lo - 1) * RW + 1];
                const int64_t rmin = recs[lo * RW];
                const int64_t dl = qKey - lmax, dr = rmin - qKey;  // distanceToRange
                center = (dl <= dr) ? (lo - 1) : lo;
            }
        }
        // one round: Hamming(q, rep) + id range of every partition reachable with P probes
        for (int l = gl; l < nd; l += G) {
            const int part = center - (P - 1) + l;
            int dd = 0, b0 = 0, sz = 0;
            if (part >= 0 && part < tb.nparts) {
                const int64_t* rec = recs + part * RW;
                dd = ham_words(qc, reinterpret_cast<const uint64_t*>(rec + 2), W);
                const int64_t os = rec[2 + W];
                b0 = static_cast<int32_t>(os);
                sz = static_cast<int32_t>(os >> 32);
            }
            w3[l * 3 + 0] = dd; w3[l * 3 + 1] = b0; w3[l * 3 + 2] = sz;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    PROBE_STAMP();         // window written
    const int np_ = act ? route_probe_replay(tb, P, nd, center, gl, w3, po) : 0;
    PROBE_STAMP();         // replay done
#ifdef FSPANN_DEBUG_STAMPS
    if (pstamp) pstamp[15] = pst_i;
#endif
#undef PROBE_STAMP
    return np_;
}

__global__ __launch_bounds__(kProbeThreads) void route_probe_kernel(RouteParams prm, int4* __restrict__ probe_out,
                                                                   int32_t* __restrict__ nprobe_out, int G) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int TD = prm.TD, W = prm.W, P = prm.P;
    const int nd = 2 * P - 1;
    const int gpb = kProbeThreads / G;                 // groups per block (G: a power of two)
    const int lgG = 31 - __clz(G);
    const int grp_in_wave = lane >> lgG, gl = lane & (G - 1);
    const int grp_in_block = tid >> lgG;
    int32_t* w3 = reinterpret_cast<int32_t*>(smem) + static_cast<size_t>(grp_in_block) * nd * 3;  // [nd][3]

    const int64_t item = static_cast<int64_t>(blockIdx.x) * gpb + grp_in_block;  // (q, td) flattened
    const int64_t nitems = prm.nq * TD;
    const bool in_range = item < nitems;
    const int64_t qi = in_range ? item / TD : 0;
    const int td = in_range ? static_cast<int>(item - qi * TD) : 0;
    const RouteTable tb = prm.tables[td];
    const uint64_t* qc = prm.codes + (qi * TD + td) * W;
    const int np = route_probe_table(prm, in_range, qc, tb, G, gl, grp_in_wave, w3, probe_out + item * P);
    if (in_range && gl == 0) nprobe_out[item] = np;
}

// ------------------------------------------------------------------------------------------
// Kernel 2: stage ids, dedupe, Java order, select.  One workgroup per query.
// ------------------------------------------------------------------------------------------
// One query, one workgroup of kThreads threads.  `smem` = the workgroup's dynamic LDS; `block_id` selects this workgroup's
// slice of the global fallback arenas (g_scratch / g_sort); probe_q [TD*P] / nprobe_q [TD] = the probe lists of THIS query
// (global memory).  Every thread of the workgroup must call (barriers inside).  Also called from the bounded select
// (route_lazy.hip.h) for a query it cannot hold: kLds = false there, so only the small arrays live in LDS.
template <bool kLds, int kThreads>
__device__ __forceinline__ void route_select_query(const RouteParams& prm, unsigned char* smem, int block_id, const int4* __restrict__ probe_q,
                                   const int32_t* __restrict__ nprobe_q, const int64_t qi) {
    int tid = threadIdx.x;                       // not const: see the register note at phase C
    constexpr int nthreads = kThreads;
    int lane = tid & 63, wave = tid >> 6;
    const int TD = prm.TD, P = prm.P, S = prm.S;
    const int TP = TD * P;

    // ---- carve the per-query arrays (LDS, or the global arena when they do not fit) -----
    unsigned char* arena = kLds ? smem : prm.g_scratch + static_cast<int64_t>(block_id) * prm.g_stride;
    size_t o = 0;
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(arena + o);  o += static_cast<size_t>(prm.sort_cap) * 8;
    uint32_t* ht = reinterpret_cast<uint32_t*>(arena + o);       o += static_cast<size_t>(prm.ht_size) * 4;
    int32_t* tup = reinterpret_cast<int32_t*>(arena + o);        o += static_cast<size_t>(prm.max_tuples) * 4;
    uint16_t* tscore = reinterpret_cast<uint16_t*>(arena + o);   o += (static_cast<size_t>(prm.max_tuples) * 2 + 15) & ~size_t(15);
    int32_t* fseq = kLds ? nullptr : reinterpret_cast<int32_t*>(arena + o);   // sliced build: tuple of the FIRST occurrence of a repeat's id
    // small arrays always live in LDS
    unsigned char* sm = kLds ? smem + o : smem;
    size_t so = 0;
    int64_t* ids_base = reinterpret_cast<int64_t*>(sm + so);     so += static_cast<size_t>(TD) * 8;
    int4* probe = reinterpret_cast<int4*>(sm + so);              so += static_cast<size_t>(TP) * 16;
    int32_t* bins = reinterpret_cast<int32_t*>(sm + so);         so += 1024 * 4;
    int32_t* duplist = reinterpret_cast<int32_t*>(sm + so);      so += kDupListMax * 4;
    int32_t* stepcnt = reinterpret_cast<int32_t*>(sm + so);      so += static_cast<size_t>(TP) * 4;
    int32_t* nprobe = reinterpret_cast<int32_t*>(sm + so);       so += static_cast<size_t>(TD) * 4;
    int32_t* dupcnt = reinterpret_cast<int32_t*>(sm + so);       // [TD]
    // Global-arena mode, hash table in LDS: the arrays indexed by tuple number (tup, tscore) are walked in order and can live in
    // global memory, the hash table is hit at random — when it fits the LDS region behind the small arrays (which phase C then
    // reuses for the ordering, as it reuses the table's space in LDS mode) it is kept there: global atomics on a 256 KB table per
    // workgroup were a third of the full select at the reference's larger profiles.
    if constexpr (!kLds) {
        if (prm.lds_sort_words >= prm.ht_size && prm.lds_sort_words > 0)
            ht = reinterpret_cast<uint32_t*>(sm + ((so + static_cast<size_t>(TD) * 4 + 15) & ~size_t(15)));
    }

    __shared__ int s_n, s_raw, s_fill, s_cut, s_star, s_need, s_b1, s_lvl1, s_ndup, s_tree, s_suspect, s_smin, s_lmax;
    __shared__ int s_slice[8];                  // sliced build: live tuples per slice of the id space
    // Sliced build (global-arena mode, long lists): the hash of ONE slice of the id space at a time, in the LDS region behind the
    // small arrays.  Random CAS / min on a 256 KB table in global memory were a third of the full select at SIFT_P10_HIGH
    // (494 us of a 1.48 ms workgroup, profiles/notes/r03_full_select_phases.txt); the tuples themselves are walked in order.
    bool sliced = false;
    uint32_t* const lds_region = reinterpret_cast<uint32_t*>(sm + ((so + static_cast<size_t>(TD) * 4 + 15) & ~size_t(15)));
    auto slice_of = [&](int32_t id) -> int { return prm.slice_bits > 0 ? static_cast<int>((static_cast<uint32_t>(id) * 0xC2B2AE35u) >> (32 - prm.slice_bits)) : 0; };

    const uint32_t ht_mask = static_cast<uint32_t>(prm.ht_size - 1);

    for (int i = tid; i < TD; i += nthreads) ids_base[i] = prm.tables[i].ids_base;

#ifdef FSPANN_DEBUG_STAMPS
#define FSP_STAMP(i) do { if (prm.dbg && tid == 0 && qi == block_id) prm.dbg[block_id * 16 + (i)] = wall_clock64(); } while (0)
#else
#define FSP_STAMP(i) do { } while (0)
#endif
#define FSP_TS(j) ((prm.S_shift >= 0) ? ((j) >> prm.S_shift) : ((j) / S))

    // hash entries are (tag(id) << seq_bits) | seq: a failed CAS can tell "other id" from the returned word alone;
    // tup[] is consulted only when the tags agree (true repeats, or a 2^-(32-seq_bits) false match)
    // floor(n / P), floor(n / SP) for n * divisor < 2^32 by one multiply-high (magic = floor(2^32 / divisor) + 1): the probes per table
    // and the pieces per partition are launch constants only the host knows, and a 32-bit division is ~35 instructions on this
    // machine — three of them per staged row
    const uint32_t magic_P = (P > 1) ? 0xFFFFFFFFu / static_cast<uint32_t>(P) + 1u : 0u;
    // (global-arena mode only — the long lists, where staging is 35 rows per wave; the LDS-mode kernel keeps the plain division: with the
    // cheap one the compiler keeps twelve rows' coordinates alive across the loads and the kernel spills)
    auto div_P = [&](int n) -> int { if constexpr (kLds) return n / P; else return (P > 1) ? static_cast<int>(__umulhi(static_cast<uint32_t>(n), magic_P)) : n; };
    const uint32_t seq_mask = (1u << prm.seq_bits) - 1u;
    const uint32_t tag_max = (0xFFFFFFFFu >> prm.seq_bits) - 1u;   // keeps every entry != kHtEmpty
    auto id_tag = [&](int32_t id) -> uint32_t {
        const uint32_t t = (static_cast<uint32_t>(id) * 0x9E3779B1u) >> prm.seq_bits;
        return t > tag_max ? tag_max : t;
    };
    // probe the hash for `id`; returns the slot that holds it (must exist)
    auto find_slot = [&](int32_t id) -> uint32_t {
        uint32_t slot = (static_cast<uint32_t>(id) * 2654435761u) >> prm.ht_shift;
        const uint32_t stp = ((static_cast<uint32_t>(id) * 0x85EBCA6Bu) >> prm.ht_shift) | 1u;
        // the id was inserted by B1, so its slot is met on its probe sequence; the trip bound (an odd step visits every
        // slot of the power-of-two table once) only keeps a host-side slip from turning into waves that never finish
        for (int tries = 0; tries < prm.ht_size; tries++) {
            const uint32_t cur = ht[slot];
            if (cur != kHtEmpty && (cur >> prm.seq_bits) == id_tag(id) && tup[cur & seq_mask] == id) return slot;
            slot = (slot + stp) & ht_mask;
        }
        return slot;
    };

    {
        FSP_STAMP(0);
        // ---- reset (16-byte stores) + probe list of this query -------------------------------
        {
            uint4* h4 = reinterpret_cast<uint4*>(ht);
            const uint4 e4 = make_uint4(kHtEmpty, kHtEmpty, kHtEmpty, kHtEmpty);
            for (int i = tid; i < prm.ht_size / 4; i += nthreads) h4[i] = e4;
            uint4* t4 = reinterpret_cast<uint4*>(tscore);
            const int nt4 = (prm.max_tuples * 2 + 15) / 16;
            for (int i = tid; i < nt4; i += nthreads) t4[i] = make_uint4(0, 0, 0, 0);
            for (int i = tid; i < 1024; i += nthreads) bins[i] = 0;
            for (int i = tid; i < TP; i += nthreads) { stepcnt[i] = 0; probe[i] = probe_q[i]; }
            for (int i = tid; i < TD; i += nthreads) { dupcnt[i] = 0; nprobe[i] = nprobe_q[i]; }
            if (tid == 0) { s_n = 0; s_raw = 0; s_fill = 0; s_cut = 0x7FFFFFFF; s_ndup = 0; s_lvl1 = 0; s_tree = 0; s_suspect = 0; }
            if (tid < 8) s_slice[tid] = 0;
        }
        __syncthreads();
        FSP_STAMP(1);

        // ---- A2: stage ids.  One wave-iteration = 64 consecutive positions of ONE probed partition, so all the
        // index arithmetic (table, step, id range, score) is wave-uniform (scalar) and the id load is one
        // coalesced 256-byte row; kStageU iterations are in flight per wave.
        {
            const int SP = (S + 63) >> 6;                 // 64-lane pieces per partition
            const uint32_t magic_SP = (SP > 1) ? 0xFFFFFFFFu / static_cast<uint32_t>(SP) + 1u : 0u;
            auto div_SP = [&](int n) -> int { if constexpr (kLds) return n / SP; else return (SP > 1) ? static_cast<int>(__umulhi(static_cast<uint32_t>(n), magic_SP)) : n; };
            const int nitems = TP * SP;
            const int nwv = nthreads >> 6;
            for (int it0 = wave; it0 < nitems; it0 += nwv * kStageU) {
                int32_t idv[kStageU];
#pragma unroll
                for (int u = 0; u < kStageU; u++) {
                    const int it = it0 + u * nwv;
                    idv[u] = -1;
                    if (it < nitems) {
                        const int ts = div_SP(it), pos = (it - ts * SP) * 64 + lane;
                        const int td = div_P(ts), step = ts - td * P;
                        const int4 pr = probe[ts];
                        if (step < nprobe[td] && pos < pr.w && pos < S) idv[u] = prm.ids[ids_base[td] + pr.z + pos];
                    }
                }
#pragma unroll
                for (int u = 0; u < kStageU; u++) {
                    const int it = it0 + u * nwv;
                    if (it >= nitems) continue;           // wave-uniform
                    const int ts = div_SP(it), pos = (it - ts * SP) * 64 + lane;
                    if (pos >= S) continue;
                    const int j = ts * S + pos;
                    int32_t id = idv[u];
                    if (id >= 0 && prm.deleted_bits && ((prm.deleted_bits[id >> 5] >> (id & 31)) & 1u)) id = -1;
                    const int sc = probe[ts].y;
                    tup[j] = id;
                    if (id >= 0) tscore[j] = static_cast<uint16_t>(sc) | kFirstFlag | kLiveFlag;
                    // histogram of (presumed) first occurrences + live tuples per probe step: one atomic per wave
                    const int c1 = __popcll(__ballot(id >= 0));
                    if (lane == 0 && c1) { atomicAdd(&bins[sc], c1); atomicAdd(&stepcnt[ts], c1); }
                    if constexpr (!kLds) {
                        if (prm.slice_ht > 0 && prm.slice_bits > 0) {
                            for (int sl = 0; sl < (1 << prm.slice_bits); sl++) {     // wave-uniform trips
                                const int cs = __popcll(__ballot(id >= 0 && slice_of(id) == sl));
                                if (lane == 0 && cs) atomicAdd(&s_slice[sl], cs);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        FSP_STAMP(8);
        if constexpr (!kLds) {
            if (prm.slice_ht > 0) {
                // every slice must leave room in its table (ids are spread by a multiplicative hash: a slice beyond 7/8 of the
                // table means crafted ids -> the arena table below)
                sliced = true;
                for (int sl = 0; sl < (1 << max(prm.slice_bits, 0)); sl++) sliced = sliced && (prm.slice_bits <= 0 || s_slice[sl] <= prm.slice_ht - (prm.slice_ht >> 3));
                if (prm.slice_bits <= 0) sliced = prm.max_tuples <= prm.slice_ht - (prm.slice_ht >> 3);
            }
            if (sliced) {
                const int hs = prm.slice_ht;
                const uint32_t hmask = static_cast<uint32_t>(hs - 1);
                const int hshift = 32 - (31 - __clz(hs));
                uint32_t* sht = lds_region;
                for (int sl = 0; sl < (1 << max(prm.slice_bits, 0)); sl++) {
                    {
                        uint4* h4 = reinterpret_cast<uint4*>(sht);
                        const uint4 e4 = make_uint4(kHtEmpty, kHtEmpty, kHtEmpty, kHtEmpty);
                        for (int i = tid; i < hs / 4; i += nthreads) h4[i] = e4;
                    }
                    __syncthreads();
                    // pass 1: sht[slot of id] = min seq over the tuples holding id (tag | seq entries, as below).  The ids of kSlU trips
                    // are requested together: one global round trip per kSlU tuples of a lane instead of one per tuple (one workgroup
                    // per CU here: nothing else hides that latency)
                    constexpr int kSlU = 8;
                    for (int j0 = tid; j0 < prm.max_tuples; j0 += nthreads * kSlU) {
                      int32_t idb[kSlU];
#pragma unroll
                      for (int u = 0; u < kSlU; u++) { const int j = j0 + u * nthreads; idb[u] = (j < prm.max_tuples) ? tup[j] : -1; }
#pragma unroll
                      for (int u = 0; u < kSlU; u++) {
                        const int j = j0 + u * nthreads;
                        const int32_t id = idb[u];
                        if (id < 0 || slice_of(id) != sl) continue;
                        uint32_t slot = (static_cast<uint32_t>(id) * 2654435761u) >> hshift;
                        const uint32_t stp = ((static_cast<uint32_t>(id) * 0x85EBCA6Bu) >> hshift) | 1u;
                        const uint32_t mytag = id_tag(id);
                        const uint32_t val = (mytag << prm.seq_bits) | static_cast<uint32_t>(j);
                        for (int tries = 0; tries < hs; tries++) {       // the table has room (checked above): an empty slot or the id is met
                            const uint32_t cur = atomicCAS(&sht[slot], kHtEmpty, val);
                            if (cur == kHtEmpty) break;
                            if ((cur >> prm.seq_bits) == mytag && tup[cur & seq_mask] == id) { atomicMin(&sht[slot], val); break; }
                            slot = (slot + stp) & hmask;
                        }
                      }
                    }
                    __syncthreads();
                    if (sl == 0) FSP_STAMP(12);
                    // pass 2: every tuple of the slice looks its id up: the entry holds the FIRST tuple of the id; any other is a repeat
                    for (int j0 = tid; j0 < prm.max_tuples; j0 += nthreads * kSlU) {
                      int32_t idb[kSlU];
#pragma unroll
                      for (int u = 0; u < kSlU; u++) { const int j = j0 + u * nthreads; idb[u] = (j < prm.max_tuples) ? tup[j] : -1; }
#pragma unroll
                      for (int u = 0; u < kSlU; u++) {
                        const int j = j0 + u * nthreads;
                        const int32_t id = idb[u];
                        if (id < 0 || slice_of(id) != sl) continue;
                        uint32_t slot = (static_cast<uint32_t>(id) * 2654435761u) >> hshift;
                        const uint32_t stp = ((static_cast<uint32_t>(id) * 0x85EBCA6Bu) >> hshift) | 1u;
                        const uint32_t mytag = id_tag(id);
                        int f = j;
                        for (int tries = 0; tries < hs; tries++) {
                            const uint32_t cur = sht[slot];
                            if (cur != kHtEmpty && (cur >> prm.seq_bits) == mytag) {
                                const int cs = static_cast<int>(cur & seq_mask);
                                if (cs == j) break;                              // the entry IS this tuple: a first occurrence — four of five
                                if (tup[cs] == id) { f = cs; break; }            //   tuples end here, without the confirming read of the arena
                            }
                            slot = (slot + stp) & hmask;
                        }
                        if (f != j) {
                            const int ts = FSP_TS(j);
                            const uint16_t v = static_cast<uint16_t>(probe[ts].y) | kLiveFlag;     // what staging wrote, less the first flag: no read
                            tscore[j] = v;
                            fseq[j] = f;
                            atomicSub(&bins[v & 0x3FFF], 1);
                            atomicSub(&stepcnt[ts], 1);
                            atomicAdd(&dupcnt[div_P(ts)], 1);
                        }
                      }
                    }
                    __syncthreads();
                    if (sl == 0) FSP_STAMP(13);
                }
            }
        }
        // ---- B1: hash build, ht[slot] = min seq of the id owning the slot.  Every lane walks ITS OWN queue of
        // tuples (j = tid, tid + nthreads, ...) one probe step per loop trip, so a lane with a long probe
        // sequence does not stall the other 63: the wave finishes after max-over-lanes of the SUM of probe
        // lengths instead of the sum of per-tuple maxima.
        if (!sliced) {
            int j = tid;
            int32_t id = -1;
            uint32_t slot = 0, stp = 1, mytag = 0;
            bool have = false;
            while (true) {
                if (!have) {
                    while (j < prm.max_tuples && (id = tup[j]) < 0) j += nthreads;
                    if (j >= prm.max_tuples) break;
                    slot = (static_cast<uint32_t>(id) * 2654435761u) >> prm.ht_shift;
                    stp = ((static_cast<uint32_t>(id) * 0x85EBCA6Bu) >> prm.ht_shift) | 1u;
                    mytag = id_tag(id);
                    have = true;
                }
                const uint32_t val = (mytag << prm.seq_bits) | static_cast<uint32_t>(j);
                const uint32_t cur = atomicCAS(&ht[slot], kHtEmpty, val);
                if (cur == kHtEmpty) {                 // created
                    have = false; j += nthreads;
                } else if ((cur >> prm.seq_bits) == mytag && tup[cur & seq_mask] == id) {  // same id
                    const uint32_t old = atomicMin(&ht[slot], val);        // equal tags: ordered by seq
                    const uint32_t loser = max(old & seq_mask, static_cast<uint32_t>(j));  // exactly one repeat per meeting
                    const int pos = atomicAdd(&s_ndup, 1);
                    if (pos < kDupListMax) duplist[pos] = static_cast<int32_t>(loser);
                    have = false; j += nthreads;
                } else {
                    slot = (slot + stp) & ht_mask;
                }
            }
        }
        __syncthreads();
        FSP_STAMP(2);

        const int ndup = s_ndup;
        // ---- repeats: drop them from "first" bookkeeping -----------------------------------------
        // (list path when they fit; otherwise recompute the flags from the hash slots; the sliced build has done it already)
        if (sliced) {
        } else if (ndup <= kDupListMax) {
            for (int l = tid; l < ndup; l += nthreads) {
                const int j = duplist[l];
                const uint16_t v = tscore[j];
                tscore[j] = v & ~kFirstFlag;
                const int ts = FSP_TS(j);
                atomicSub(&bins[v & 0x3FFF], 1);
                atomicSub(&stepcnt[ts], 1);
                atomicAdd(&dupcnt[div_P(ts)], 1);
            }
        } else {
            for (int j = tid; j < prm.max_tuples; j += nthreads) tscore[j] &= ~kFirstFlag;
            for (int i = tid; i < 1024; i += nthreads) bins[i] = 0;
            for (int i = tid; i < TP; i += nthreads) stepcnt[i] = 0;
            __syncthreads();
            for (int sl = tid; sl < prm.ht_size; sl += nthreads) {
                const uint32_t hv = ht[sl];
                const uint32_t j = hv & seq_mask;
                if (hv == kHtEmpty) continue;
                tscore[j] |= kFirstFlag;
                atomicAdd(&bins[tscore[j] & 0x3FFF], 1);
                atomicAdd(&stepcnt[FSP_TS(static_cast<int>(j))], 1);
            }
            __syncthreads();
            for (int j = tid; j < prm.max_tuples; j += nthreads) {
                const uint16_t v = tscore[j];
                if ((v & kLiveFlag) && !(v & kFirstFlag)) atomicAdd(&dupcnt[div_P(FSP_TS(j))], 1);
            }
        }
        __syncthreads();

        // ---- B2: HARD_CAP — first probe step at which bestScore.size() reaches the cap ----------------
        int cut = 0x7FFFFFFF;
        if (prm.need_cap) {
            if (wave == 0) {      // running sum of the steps' new ids, 64 steps per trip (one thread adding them up: 16 us at 560 steps)
                int carry = 0, c = 0x7FFFFFFF;
                for (int g0 = 0; g0 < TP; g0 += 64) {        // wave-uniform trips
                    const int g = g0 + lane;
                    int incl = (g < TP) ? stepcnt[g] : 0;
                    for (int off = 1; off < 64; off <<= 1) { const int u2 = __shfl_up(incl, off); if (lane >= off) incl += u2; }
                    const unsigned long long hit = __ballot(g < TP && carry + incl >= prm.hard_cap);
                    if (hit) { c = g0 + __ffsll(static_cast<long long>(hit)) - 1; break; }  // steps after c never run (PIS:657-659)
                    carry += __shfl(incl, 63);
                }
                if (lane == 0) s_cut = c;
            }
            __syncthreads();
            cut = s_cut;
            if (cut != 0x7FFFFFFF) {  // entries and repeats behind the cut never existed
                for (int j = (cut + 1) * S + tid; j < prm.max_tuples; j += nthreads) {      // (tuple j belongs to step j / S)
                    const uint16_t v = tscore[j];
                    if (!(v & kLiveFlag) || FSP_TS(j) <= cut) continue;
                    if (v & kFirstFlag) atomicSub(&bins[v & 0x3FFF], 1);
                    else atomicSub(&dupcnt[div_P(FSP_TS(j))], 1);
                    tscore[j] = 0;
                }
                __syncthreads();
            }
        }
        // n = number of live first occurrences
        {
            int c = 0;
            for (int g = tid; g < TP; g += nthreads) if (g <= cut) c += stepcnt[g];
            if (c) atomicAdd(&s_n, c);
        }
        FSP_STAMP(9);

        // ---- B3: repeated occurrences: min score + strict improvements in table order (PIS:744-750) --
        bool b3_done = false;
        if (!kLds && sliced && !prm.out_raw) {
            // Nobody asked for rawSeen (the search path never does: it is a profiler counter of the reference, PIS:744-750), so all
            // that is needed is every id's LOWEST score — no order among its repeats, no table phases: the live repeats are listed
            // once (LDS region, free between the hash build and the ordering: (score | tuple, first occurrence) per repeat), every
            // one takes a 32-bit minimum on its first occurrence's word of fseq (unused for a first occurrence), and the repeat that
            // finds ITS OWN (score | tuple) there afterwards is the id's unique lowest: it rewrites the entry if it improves it.
            // SIFT_P10_HIGH: 56 table phases of three dependent arena reads each were 60 us of a 450 us query.
            if (wave == 0) {
                int c = 0;
                for (int td = lane; td < TD; td += 64) c += dupcnt[td];
                for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
                if (lane == 0) { s_b1 = c; s_fill = 0; }        // (s_b1, s_fill: free until the level cuts / the compaction)
            }
            __syncthreads();
            const int nrep = s_b1;
            if (2 * static_cast<int64_t>(nrep) <= prm.lds_sort_words) {      // uniform
                uint32_t* rl = lds_region;
                constexpr int kRlU = 8;
                for (int jb = 0; jb < prm.max_tuples; jb += nthreads * kRlU) {     // uniform trips: the ballots need every lane
                    uint16_t vb[kRlU];
                    int32_t fb[kRlU];
#pragma unroll
                    for (int u = 0; u < kRlU; u++) { const int j = jb + u * nthreads + tid; vb[u] = (j < prm.max_tuples) ? tscore[j] : static_cast<uint16_t>(0); }
#pragma unroll
                    for (int u = 0; u < kRlU; u++) {
                        const int j = jb + u * nthreads + tid;
                        const bool rep = (vb[u] & kLiveFlag) && !(vb[u] & kFirstFlag);
                        fb[u] = rep ? fseq[j] : -1;
                    }
#pragma unroll
                    for (int u = 0; u < kRlU; u++) {
                        const int j = jb + u * nthreads + tid;
                        const bool rep = fb[u] >= 0;
                        const unsigned long long bm = __ballot(rep);
                        int basepos = 0;
                        if (bm) {
                            if (lane == 0) basepos = atomicAdd(&s_fill, __popcll(bm));
                            basepos = __shfl(basepos, 0);
                        }
                        if (rep) {
                            const int pos = basepos + __popcll(bm & ((1ull << lane) - 1ull));
                            rl[2 * pos] = (static_cast<uint32_t>(vb[u] & 0x3FFFu) << 22) | static_cast<uint32_t>(j);
                            rl[2 * pos + 1] = static_cast<uint32_t>(fb[u]);
                            fseq[fb[u]] = -1;                     // = 0xFFFFFFFF: above every (score | tuple)
                        }
                    }
                }
                __threadfence_block();
                __syncthreads();
                const int nr = s_fill;
                for (int i = tid; i < nr; i += nthreads) atomicMin(reinterpret_cast<uint32_t*>(fseq) + rl[2 * i + 1], rl[2 * i]);
                __threadfence_block();
                __syncthreads();
                for (int i = tid; i < nr; i += nthreads) {
                    const uint32_t mine = rl[2 * i];
                    const int f = static_cast<int>(rl[2 * i + 1]);
                    // (an atomic load: the minimum was taken in L2, a plain load may be served by this CU's L1 copy of the -1 stored above)
                    const uint32_t low = __hip_atomic_load(reinterpret_cast<uint32_t*>(fseq) + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (low != mine) continue;
                    const int sc = static_cast<int>(mine >> 22);
                    const int first_sc = probe[FSP_TS(f)].y;
                    if (sc < first_sc) {
                        tscore[f] = static_cast<uint16_t>(sc) | kFirstFlag | kLiveFlag;
                        atomicSub(&bins[first_sc], 1);
                        atomicAdd(&bins[sc], 1);
                    }
                }
                __syncthreads();
                b3_done = true;
            }
            if (tid == 0) { s_fill = 0; s_b1 = 0; }
            __syncthreads();
        }
        if (b3_done) {
        } else if (sliced) {
            for (int td = 0; td < TD; td++) {   // table phases (an id occurs at most once per table); the first occurrence comes from fseq
                if (dupcnt[td] == 0) continue;  // uniform
                int improved = 0;
                const int j1 = (td * P + nprobe[td]) * S;
                for (int j = td * P * S + tid; j < j1; j += nthreads) {
                    const uint16_t v = tscore[j];
                    if (!(v & kLiveFlag) || (v & kFirstFlag)) continue;
                    const int f = fseq[j];
                    const int sc = probe[FSP_TS(j)].y;
                    const int old = tscore[f] & 0x3FFF;
                    if (sc < old) {
                        tscore[f] = static_cast<uint16_t>(sc) | kFirstFlag | kLiveFlag;
                        atomicSub(&bins[old], 1);
                        atomicAdd(&bins[sc], 1);
                        improved++;
                    }
                }
                if (improved) atomicAdd(&s_raw, improved);
                __syncthreads();
            }
        } else if (ndup > 0 && ndup <= kDupListMax && prm.nbins <= 256 && prm.ht_size <= 65536) {
            // All repeats in parallel: a repeat improves iff its score is below the first occurrence's and below every earlier
            // repeat of the same id; the overall minimum (earliest on ties) is the one that rewrites the entry's score.
            // "The same id" = "the same hash slot", and the slot doubles as a counter: once every repeat has found its slot, each
            // live one adds 1 to the slot's TAG field (nothing reads the table after this step), so (tag now - tag of the id) is the
            // number of live repeats of that id.  A lone repeat — most of them — is settled on the spot; the others (a few dozen
            // to a few hundred) are listed and walk only that list.  (Every repeat walking all of them, ids read from the arena:
            // 97 us of a 300 us query at SIFT_P4_FAST on SIFT-like data; with the ids in LDS: 28 us; this: 9.5 us.)
            // Scratch in bins[256 ...), free here (the score histogram uses bins[0, nbins) only): the list of those with company
            // (slot | index).  duplist[l] becomes (score << 23 | live << 22 | tuple number).
            uint32_t* mlist = reinterpret_cast<uint32_t*>(bins + 256);                           // [kDupListMax] slot << 16 | index in duplist
            static_assert(256 + kDupListMax <= 1024 && kDupListMax % 8 == 0 && kDupListMax <= 65536 && kSeqBits == 22, "scratch behind the score histogram; a tuple number is 22 bits");
            constexpr int kDupTrips = (kDupListMax + kThreads - 1) / kThreads;
            uint32_t my_slot[kDupTrips], my_f[kDupTrips], my_tag[kDupTrips];
            if (tid == 0) s_fill = 0;                                  // (s_fill: free until the compaction; reset below)
#pragma unroll
            for (int t = 0; t < kDupTrips; t++) {
                const int l = tid + t * kThreads;
                my_slot[t] = 0; my_f[t] = 0; my_tag[t] = 0;
                if (l < ndup) {
                    const int j = duplist[l];
                    const int32_t id = tup[j];
                    const bool live = (tscore[j] & kLiveFlag) != 0;
                    duplist[l] = j | (live ? (1 << 22) : 0) | (probe[FSP_TS(j)].y << 23);
                    const uint32_t sl = find_slot(id);
                    my_slot[t] = sl;
                    my_f[t] = ht[sl] & seq_mask;
                    my_tag[t] = id_tag(id);
                }
            }
            __syncthreads();                                           // every repeat has its slot: the table's tags may go
#pragma unroll
            for (int t = 0; t < kDupTrips; t++) {
                const int l = tid + t * kThreads;
                if (l < ndup && (duplist[l] & (1 << 22))) atomicAdd(&ht[my_slot[t]], 1u << prm.seq_bits);
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < kDupTrips; t++) {
                const int l = tid + t * kThreads;
                if (l >= ndup) continue;
                const int me = duplist[l];
                if (!(me & (1 << 22))) continue;                       // behind the HARD_CAP cut
                const uint32_t company = ((ht[my_slot[t]] >> prm.seq_bits) - my_tag[t]) & (0xFFFFFFFFu >> prm.seq_bits);
                if (company > 1u) { mlist[atomicAdd(&s_fill, 1)] = (my_slot[t] << 16) | static_cast<uint32_t>(l); continue; }
                const int sc = static_cast<int>(static_cast<uint32_t>(me) >> 23);
                const int first_sc = probe[FSP_TS(static_cast<int>(my_f[t]))].y;
                if (sc < first_sc) {                                   // the id's only repeat: it improves, and it is the minimum
                    atomicAdd(&s_raw, 1);
                    tscore[my_f[t]] = static_cast<uint16_t>(sc) | kFirstFlag | kLiveFlag;
                    atomicSub(&bins[first_sc], 1);
                    atomicAdd(&bins[sc], 1);
                }
            }
            __syncthreads();
            const int nmulti = s_fill;
            const uint2* m2p = reinterpret_cast<const uint2*>(mlist);       // (8-byte aligned: bins is)
            for (int mi = tid; mi < nmulti; mi += nthreads) {
                const uint32_t mine = mlist[mi];
                const int l = static_cast<int>(mine & 0xFFFFu);
                const uint32_t sl = mine >> 16;
                const int me = duplist[l];
                const int j = me & static_cast<int>(kSeqMask);
                const int sc = static_cast<int>(static_cast<uint32_t>(me) >> 23);
                const uint32_t f = ht[sl] & seq_mask;                   // (the adds went to the tag field)
                const int first_sc = probe[FSP_TS(static_cast<int>(f))].y;
                bool improves = sc < first_sc, is_min = improves;
                // eight list entries per trip (four 8-byte LDS reads issued together; entries past the end are stale and are ruled out
                // by their position)
                for (int m0 = 0; m0 < nmulti && (improves || is_min); m0 += 8) {
                    const uint2 a0 = m2p[m0 / 2], a1 = m2p[m0 / 2 + 1], a2 = m2p[m0 / 2 + 2], a3 = m2p[m0 / 2 + 3];
                    const uint32_t e8[8] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y};
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if ((e8[u] >> 16) != sl || m0 + u >= nmulti || m0 + u == mi) continue;
                        const int o2 = duplist[e8[u] & 0xFFFFu];
                        const int j2 = o2 & static_cast<int>(kSeqMask);
                        const int sc2 = static_cast<int>(static_cast<uint32_t>(o2) >> 23);
                        if (j2 < j && sc2 <= sc) improves = false;
                        if (sc2 < sc || (sc2 == sc && j2 < j)) is_min = false;
                    }
                }
                if (improves) atomicAdd(&s_raw, 1);
                if (is_min) {  // unique writer per id
                    tscore[f] = static_cast<uint16_t>(sc) | kFirstFlag | kLiveFlag;
                    atomicSub(&bins[first_sc], 1);
                    atomicAdd(&bins[sc], 1);
                }
            }
            __syncthreads();
            for (int l = tid; l < kDupListMax; l += nthreads) bins[256 + l] = 0;   // bins[256 ...) back to zeros
            if (tid == 0) s_fill = 0;
            __syncthreads();
        } else if (ndup > 0) {
            for (int td = 0; td < TD; td++) {   // table phases: an id occurs at most once per table
                if (dupcnt[td] == 0) continue;  // uniform
                int improved = 0;
                const int j1 = (td * P + nprobe[td]) * S;
                for (int j = td * P * S + tid; j < j1; j += nthreads) {
                    const uint16_t v = tscore[j];
                    if (!(v & kLiveFlag) || (v & kFirstFlag)) continue;
                    const uint32_t f = ht[find_slot(tup[j])] & seq_mask;
                    const int sc = probe[FSP_TS(j)].y;
                    const int old = tscore[f] & 0x3FFF;
                    if (sc < old) {
                        tscore[f] = static_cast<uint16_t>(sc) | kFirstFlag | kLiveFlag;
                        atomicSub(&bins[old], 1);
                        atomicAdd(&bins[sc], 1);
                        improved++;
                    }
                }
                if (improved) atomicAdd(&s_raw, improved);
                __syncthreads();
            }
        } else {
            __syncthreads();
        }
        FSP_STAMP(3);
        // Register note (as in route_lazy_run): redefining the thread's coordinates here (the asm changes nothing) keeps the
        // per-thread offsets of the phases below from being computed up front and held through the phases above.
        asm volatile("" : "+v"(tid), "+v"(lane), "+v"(wave));

        // ---- C: order + select -------------------------------------------------------------
        const int n = s_n;
        const int capf = java_final_cap(prm.cap0, n);
        const int capbits = 31 - __clz(capf);
        const uint32_t bmask = static_cast<uint32_t>(capf - 1);
        const int bshift = kBucketBits - capbits;  // align the bucket's MSB with the 20-bit field
        if (wave == 0) {  // level 0: cut score from the histogram
            int star = prm.nbins - 1, need = 0x7FFFFFFF, tie = 0;
            if (n > prm.limit) {
                int before = 0;
                star = wave_find_cut(bins, prm.nbins, prm.limit, lane, &before);
                need = prm.limit - before;
                tie = bins[star];
            }
            {   // lowest occupied level and the largest level up to the cut: the long-list ordering sizes its groups by them
                int lo = 0x7FFFFFFF, big = 0;
                for (int b = lane; b <= star; b += 64) {
                    const int cb = bins[b];
                    if (cb > 0) { lo = min(lo, b); big = max(big, cb); }
                }
                for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); big = max(big, __shfl_xor(big, off)); }
                if (lane == 0) { s_smin = (lo == 0x7FFFFFFF) ? star : lo; s_lmax = big; }
            }
            if (lane == 0) {
                s_star = star;   // entries with score < star are all selected; `need` more come from score == star
                s_need = need;
                s_b1 = 0x7FFFFFFF;
                s_lvl1 = (n > prm.limit && tie - need > 256) ? 1 : 0;  // worth cutting the tie group further
            }
        }
        __syncthreads();
        const int star = s_star;
        const bool lvl1 = (s_lvl1 != 0);
        auto bucket_field = [&](int32_t id) -> uint32_t {
            uint32_t h = prm.decimal_ids ? decimal_string_hash_dev(static_cast<uint32_t>(id))
                                         : static_cast<uint32_t>(prm.java_hash[id]);
            h ^= (h >> 16);  // HashMap.hash(): spread
            return (h & bmask) << bshift;
        };
        if (lvl1) {
            // level 1: tie group (score == star) by the top 10 bits of the HashMap bucket
            for (int i = tid; i < 1024; i += nthreads) bins[i] = 0;
            __syncthreads();
            for (int j = tid; j < prm.max_tuples; j += nthreads) {
                const uint16_t v = tscore[j];
                if (!(v & kFirstFlag) || (v & 0x3FFFu) != star) continue;
                atomicAdd(&bins[bucket_field(tup[j]) >> (kBucketBits - 10)], 1);
            }
            __syncthreads();
            if (wave == 0) {
                int before = 0;
                const int b1v = wave_find_cut(bins, 1024, s_need, lane, &before);
                if (lane == 0) s_b1 = b1v;
            }
            __syncthreads();
        }
        const int b1 = s_b1;
        FSP_STAMP(11);
        // compaction of the surviving candidates: score < star, or score == star && bucket bin <= b1
        uint64_t* gs = prm.g_sort ? prm.g_sort + static_cast<int64_t>(block_id) * prm.g_sort_stride : sortbuf;
        constexpr int kCmU = 4;                                      // trips whose tuple words are requested together (global-arena mode: one
        for (int jb = 0; jb < prm.max_tuples; jb += nthreads * kCmU) {   //   round trip per kCmU trips; uniform trip counts: the ballots need every lane)
          uint16_t vb[kCmU];
          int32_t ib[kCmU];
#pragma unroll
          for (int u = 0; u < kCmU; u++) {
              const int j = jb + u * nthreads + tid;
              vb[u] = (j < prm.max_tuples) ? tscore[j] : static_cast<uint16_t>(0);
              ib[u] = (j < prm.max_tuples) ? tup[j] : -1;
          }
#pragma unroll
          for (int u = 0; u < kCmU; u++) {
            const int j0 = jb + u * nthreads;
            if (j0 >= prm.max_tuples) break;                           // uniform
            const int j = j0 + tid;
            bool take = false;
            uint64_t key = 0;
            if (j < prm.max_tuples) {
                const uint16_t v = vb[u];
                const int sc = v & 0x3FFFu;
                if ((v & kFirstFlag) && sc <= star) {
                    const uint32_t bfield = bucket_field(ib[u]);
                    if (!(sc == star && lvl1 && static_cast<int>(bfield >> (kBucketBits - 10)) > b1)) {
                        take = true;
                        key = (static_cast<uint64_t>(sc) << (kBucketBits + kSeqBits)) |
                              (static_cast<uint64_t>(bfield) << kSeqBits) | static_cast<uint32_t>(j);
                    }
                }
            }
            // one LDS atomic per wave instead of one per survivor (they would all hit the same address)
            const unsigned long long bm = __ballot(take);
            int basepos = 0;
            if (bm) {
                if (lane == 0) basepos = atomicAdd(&s_fill, __popcll(bm));
                basepos = __shfl(basepos, 0);
            }
            if (take) {
                const int pos = basepos + __popcll(bm & ((1ull << lane) - 1ull));
                if (pos < prm.sort_cap) sortbuf[pos] = key;
                else gs[pos] = key;  // only reachable when g_sort exists (host guarantees capacity)
            }
          }
        }
        __syncthreads();
        FSP_STAMP(4);
        const int nsel = s_fill;
#ifdef FSPANN_DEBUG_STAMPS
        if (tid == 0 && prm.dbg && qi == block_id) prm.dbg[block_id * 16 + 15] = nsel;
#endif
        const int nout = min(nsel, prm.limit);
        if (nsel <= kRankSortMax - 128 && prm.sort_cap >= kRankSortMax) {
            // all-pairs rank: keys are unique (seq is), so rank = #smaller keys; no barriers.
            // element i is ranked by `parts` cooperating lanes, each over a slice of j.
            // slices of 8 keys are read with four 16-byte LDS loads issued back to back.
            int parts = 1;
            while (parts < 16 && nsel * parts * 2 <= nthreads * 2) parts <<= 1;
            const int len = (((nsel + parts - 1) / parts) + 7) & ~7;   // slice length, multiple of 8
            const int padded = parts * len;                            // <= sort_cap by construction (host: sort_cap >= 1024)
            for (int i = nsel + tid; i < padded; i += nthreads) sortbuf[i] = ~0ull;  // sentinels never count
            __syncthreads();
            const int per_pass = nthreads / parts;
            for (int base = 0; base < nsel; base += per_pass) {
                const int i = base + tid / parts, part = tid % parts;
                uint64_t key = 0;
                int rank = 0;
                if (i < nsel) {
                    key = sortbuf[i];
                    const ulonglong2* sb2 = reinterpret_cast<const ulonglong2*>(sortbuf + part * len);
                    for (int j = 0; j < len / 2; j += 4) {
                        const ulonglong2 k0 = sb2[j], k1 = sb2[j + 1], k2 = sb2[j + 2], k3 = sb2[j + 3];
                        rank += (k0.x < key) + (k0.y < key) + (k1.x < key) + (k1.y < key) +
                                (k2.x < key) + (k2.y < key) + (k3.x < key) + (k3.y < key);
                    }
                }
                for (int off = parts >> 1; off > 0; off >>= 1) rank += __shfl_xor(rank, off);
                if (i < nsel && part == 0 && rank < nout) {
                    prm.out_ids[qi * prm.out_cap + rank] = tup[static_cast<uint32_t>(key) & kSeqMask];
                    if (prm.out_score) prm.out_score[qi * prm.out_cap + rank] = static_cast<int32_t>(key >> (kBucketBits + kSeqBits));
                }
            }
        } else if (prm.g_sub && capbits + prm.seq_bits <= 32 && prm.nbins <= 512 &&
                   (kLds ? prm.ht_size : prm.lds_sort_words) >= 4096) {
            // ---- long lists (the reference's shipped profiles: 6-28 k entries): the order key is score | bin | first seq.  There are
            // at most `bits + 1` scores and the bins are spread evenly, so the list is cut into GROUPS by (score, top bits of the bin)
            // — a histogram, a prefix sum and ONE scatter pass of 32-bit sub-keys (bin | seq) — and every group (tens to hundreds of
            // entries) is sorted by ONE WAVE on its own LDS slice, the workgroup's waves taking groups off a counter: no workgroup
            // barrier inside the sort.  (A bitonic sort of the whole list in global memory — the general fall-back below — was 76 % of
            // the full select at SIFT_P4_FAST and 68 % at SIFT_P10_HIGH, tools/route_full_stamps.py.)
            uint32_t* gsort = kLds ? ht : reinterpret_cast<uint32_t*>(sm + ((so + static_cast<size_t>(TD) * 4 + 15) & ~size_t(15)));
            const int gcap_all = kLds ? prm.ht_size : prm.lds_sort_words;
            // Groups = (score level, top gb bits of the bin).  Only the levels smin .. star can hold a selected entry, and the LARGEST of
            // them decides how finely the bins must be cut for a group to fit a wave's slice: gb = enough bits to bring that level
            // down to ~128 entries per group (ids spread evenly over the bins), as far as 1 024 groups allow.  (With a fixed four or
            // five bits over all `bits + 1` levels, data whose scores crowd on two or three levels — SIFT-like vectors — filled groups
            // beyond a wave's slice, and those are sorted by the whole workgroup one after the other: up to 240 us of a query.)
            const int smin = min(s_smin, star), nlev = star - smin + 1;
            int gb = 0;                                      // top bits of the bin that join the score in the group number
            // (~128 per group: with 64 the sort gains a tenth, with 32 the per-group overhead takes it back — tools/route_full_stamps.py)
            while (gb < 10 && gb < capbits && (nlev << (gb + 1)) <= 1024 && (s_lmax >> gb) > 128) gb++;
            const int ngrp = nlev << gb;
#ifdef FSPANN_DEBUG_STAMPS
            if (tid == 0 && prm.dbg && qi == block_id) prm.dbg[block_id * 16 + 15] = nsel + (static_cast<long long>(ngrp) << 20) + (static_cast<long long>(nlev) << 32) + (static_cast<long long>(s_lmax) << 40);
#endif
            int32_t* cursor = reinterpret_cast<int32_t*>(gsort) + (gcap_all - 1024);      // [ngrp <= 1024] running output position per group
            const int nwv = nthreads >> 6;
            // LDS region: [per-wave slices: nwv * wcap][the sub-keys, when they fit][cursors: 1024].  With the sub-keys in LDS a group
            // costs no global round trip (one per group and wave was most of what was left of the ordering at SIFT_P4_FAST).
            int wcap = 64;                                   // entries one wave can sort in its slice (a power of two)
            const bool sub_lds = nwv * 64 + nsel <= gcap_all - 1024;
            const int room = sub_lds ? ((gcap_all - 1024 - nsel) & ~3) : gcap_all - 1024;
            while (wcap * 2 * nwv <= room && wcap < 4096) wcap <<= 1;
            if (!prm.wave_sort) wcap = 0;
            uint32_t* gsub = sub_lds ? gsort + ((gcap_all - 1024 - nsel) & ~3) : prm.g_sub + static_cast<int64_t>(block_id) * prm.g_sub_stride;
            __syncthreads();                                 // ht, bins: every wave is past their last use
            for (int i = tid; i < ngrp; i += nthreads) bins[i] = 0;
            if (tid == 0) s_cut = 0;                         // s_cut: a group did not fit a wave's slice
            __syncthreads();
            auto key_at = [&](int i) -> uint64_t { return (i < prm.sort_cap) ? sortbuf[i] : gs[i]; };
            auto group_of = [&](uint64_t key) -> int {
                const int sc = static_cast<int>(key >> (kBucketBits + kSeqBits));
                const uint32_t bucket = (static_cast<uint32_t>(key >> kSeqBits) & ((1u << kBucketBits) - 1u)) >> bshift;
                return ((sc - smin) << gb) | static_cast<int>(bucket >> (capbits - gb));
            };
            for (int i = tid; i < nsel; i += nthreads) atomicAdd(&bins[group_of(key_at(i))], 1);
            __syncthreads();
            if (wave == 0) {                                 // exclusive prefix of the group sizes
                int carry = 0;
                for (int g0 = 0; g0 < ngrp; g0 += 64) {
                    const int g = g0 + lane;
                    const int v = (g < ngrp) ? bins[g] : 0;
                    int incl = v;
                    for (int off = 1; off < 64; off <<= 1) { const int u2 = __shfl_up(incl, off); if (lane >= off) incl += u2; }
                    if (g < ngrp) cursor[g] = carry + incl - v;
                    carry += __shfl(incl, 63);
                }
            }
            __syncthreads();
            FSP_STAMP(7);
            for (int i = tid; i < nsel; i += nthreads) {     // scatter: group by group (any order inside a group: it is sorted next)
                const uint64_t key = key_at(i);
                const uint32_t bucket = (static_cast<uint32_t>(key >> kSeqBits) & ((1u << kBucketBits) - 1u)) >> bshift;
                const uint32_t sub = (bucket << prm.seq_bits) | (static_cast<uint32_t>(key) & kSeqMask);
                gsub[atomicAdd(&cursor[group_of(key)], 1)] = sub;
            }
            __threadfence_block();
            __syncthreads();                                 // cursor[g] is now the END of group g; its start = end - bins[g]
            FSP_STAMP(10);
            const uint32_t seqm = (1u << prm.seq_bits) - 1u;
            uint32_t* slice = gsort + static_cast<size_t>(wave) * wcap;
            // Every wave on its own: wave w takes the groups w, w + waves, ... (ascending, so it may stop at the first one behind the
            // limit).  (Groups handed out through an LDS counter — `if (lane == 0) g = atomicAdd(..)` + broadcast inside a loop with
            // continue / break — hung the GPU.  The ISA of the reduced case, tools/micro/lane0_loop.hip: the structurised loop takes
            // lanes out of EXEC one by one at its latch (s_andn2_b64 exec, exec, <left>), and the broadcast is a ds_bpermute from
            // lane 0, which reads 0 from a lane that is not in EXEC — so as soon as lane 0 has left and another lane has not, the
            // rest re-run group 0 for ever.  The idiom needs every exit to be wave-uniform at run time; the variant that hung had
            // a per-lane exit in front of the broadcast.  The remaining lane-0-atomic + broadcast sites — the compaction above and two
            // in route_lazy.hip.h — sit in loops whose trip counts come from scalar / LDS values behind a barrier.)
            const int wave_s = __builtin_amdgcn_readfirstlane(wave);
            bool any_big = false;
            for (int g = wave_s; g < ngrp; g += nwv) {
                const int c = __builtin_amdgcn_readfirstlane(bins[g]);
                const int g0 = __builtin_amdgcn_readfirstlane(cursor[g]) - c;
                if (c > 0 && g0 >= nout) break;              // this group and every later one lie behind the limit
                if (c > wcap) any_big = true;
                if (c == 0 || c > wcap) continue;
                // the group's sub-keys are unique (they carry the seq), so an element's place is the number of smaller ones: an
                // all-pairs count over the slice — broadcast 16-byte reads, independent of each other — instead of a sorting network
                // whose every stage waits for the LDS (a lone wave: ~100 cycles per stage, 21 stages for 64 elements)
                const int c4 = (c + 3) & ~3;
                for (int i = lane; i < c4; i += 64) slice[i] = (i < c) ? gsub[g0 + i] : 0xFFFFFFFFu;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int scg = (g >> gb) + smin;
                const uint4* s4 = reinterpret_cast<const uint4*>(slice);
                for (int i = lane; i < c; i += 64) {
                    const uint32_t my = slice[i];
                    int rk = 0;
                    for (int j = 0; j < c4 / 4; j++) {
                        const uint4 v = s4[j];
                        rk += (v.x < my) + (v.y < my) + (v.z < my) + (v.w < my);
                    }
                    const int rank = g0 + rk;
                    if (rank < nout) {
                        prm.out_ids[qi * prm.out_cap + rank] = tup[my & seqm];
                        if (prm.out_score) prm.out_score[qi * prm.out_cap + rank] = scg;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();             // the slice is reused by this wave's next group
            }
            if (any_big && lane == 0) s_cut = 1;
            __syncthreads();
            FSP_STAMP(14);
            bool too_big = false;
            if (s_cut) {                                     // groups beyond a wave's slice: the whole workgroup sorts them one by one
                for (int g = 0; g < ngrp; g++) {             // block-uniform
                    const int c = bins[g];
                    if (c <= wcap) continue;
                    const int g0 = cursor[g] - c;
                    if (g0 >= nout) break;
                    int n2 = 1;
                    while (n2 < c) n2 <<= 1;
                    if (n2 > room) { too_big = true; break; }                // the PADDED group must fit in front of the sub-keys / cursors
                    for (int i = tid; i < n2; i += nthreads) gsort[i] = (i < c) ? gsub[g0 + i] : 0xFFFFFFFFu;
                    __syncthreads();
                    bitonic_sort_u32(gsort, n2, tid, nthreads);
                    const int scg = (g >> gb) + smin;
                    for (int i = tid; i < c; i += nthreads) {
                        const int rank = g0 + i;
                        if (rank < nout) {
                            prm.out_ids[qi * prm.out_cap + rank] = tup[gsort[i] & seqm];
                            if (prm.out_score) prm.out_score[qi * prm.out_cap + rank] = scg;
                        }
                    }
                    __syncthreads();
                }
            }
            if (too_big) {                                   // rare: redo with the general path (the keys are still where the compaction put them)
                int n2 = 1;
                while (n2 < nsel) n2 <<= 1;
                for (int i = tid; i < min(nsel, prm.sort_cap); i += nthreads) gs[i] = sortbuf[i];
                for (int i = nsel + tid; i < n2; i += nthreads) gs[i] = ~0ull;
                __syncthreads();
                bitonic_sort_u64(gs, n2, tid, nthreads);
                for (int i = tid; i < nout; i += nthreads) {
                    const uint64_t key = gs[i];
                    prm.out_ids[qi * prm.out_cap + i] = tup[static_cast<uint32_t>(key) & kSeqMask];
                    if (prm.out_score) prm.out_score[qi * prm.out_cap + i] = static_cast<int32_t>(key >> (kBucketBits + kSeqBits));
                }
            }
        } else {
            int n2 = 1;
            while (n2 < nsel) n2 <<= 1;
            const bool in_lds_buf = (n2 <= prm.sort_cap);
            if (!in_lds_buf) {  // gather everything into the global buffer
                for (int i = tid; i < min(nsel, prm.sort_cap); i += nthreads) gs[i] = sortbuf[i];
            }
            uint64_t* sb = in_lds_buf ? sortbuf : gs;
            for (int i = nsel + tid; i < n2; i += nthreads) sb[i] = ~0ull;
            __syncthreads();
            if (in_lds_buf) bitonic_sort_u64(sortbuf, n2, tid, nthreads);
            else bitonic_sort_u64(gs, n2, tid, nthreads);
            for (int i = tid; i < nout; i += nthreads) {
                const uint64_t key = sb[i];
                prm.out_ids[qi * prm.out_cap + i] = tup[static_cast<uint32_t>(key) & kSeqMask];
                if (prm.out_score) prm.out_score[qi * prm.out_cap + i] = static_cast<int32_t>(key >> (kBucketBits + kSeqBits));
            }
        }
        FSP_STAMP(5);
        // ---- T: would java.util.HashMap have treeified a bin?  The order key above (bucket, first insertion) is the
        // iteration order of `bestScore` (PIS:619,690-693) only while every bin is a plain chain.  putVal() turns a bin into a
        // red-black tree when a put finds 8 nodes in it (table >= 64, guaranteed by cap0 >= 64): from then on the bin's
        // iteration order is not insertion order and is NOT modelled here.  Detect it exactly and flag the query
        // (out_count = -1, prm.unmodelled++) instead of returning a list the JVM would not produce.
        //   stage k of the map: table length cap0 << k while size <= thr_k (thr_0 = 0.75 cap0, doubling); a bin of stage k
        //   treeifies iff >= 9 of the first thr_k + 1 distinct ids (in insertion order) share it.
        // ht (dead since B3) is reused: first as direct-indexed counters over the cap0 buckets folded to ht_size (a cheap
        // necessary condition: bins only split when the table grows), then — only for a suspect query, or when the map
        // resized — as an open-addressed (bucket -> count) table per stage.
        {
            const int thr0 = static_cast<int>(static_cast<float>(prm.cap0) * 0.75f);
            const bool single = (n <= thr0);                       // the map never resized: one stage, no ranks needed
            auto spread_of = [&](int32_t id) -> uint32_t {
                uint32_t h = prm.decimal_ids ? decimal_string_hash_dev(static_cast<uint32_t>(id)) : static_cast<uint32_t>(prm.java_hash[id]);
                return h ^ (h >> 16);
            };
            __syncthreads();       // ht, tscore: every wave is past its last use
            // Direct 4-bit counters, one per bin of the stage's table, in LDS (cap / 8 words: 16 KB for 32 768 bins): an increment
            // that finds eight is the ninth node of its bin — exact, no hash, no second look.  (A counter passes nine before it can
            // wrap, and a wrap only carries into its neighbour, so nothing is flagged that has not been flagged already.)  This is what
            // the long lists of the shipped profiles use — bestScore resizes there (n > 0.75 cap0), and the (bin -> count) hash in the
            // global arena was 228 us of a 1.48 ms workgroup at SIFT_P10_HIGH.
            uint32_t* nib = kLds ? ht : lds_region;
            const int nib_words = kLds ? prm.ht_size : prm.lds_sort_words;
            const bool use_nib = capf >= 64 && (capf >> 3) <= nib_words && !(prm.dev_flags & 2);
            if (use_nib) {
                if (!single && wave == 0) {     // exclusive prefix of stepcnt over the steps that ran (insertion ranks, as below)
                    int carry = 0;
                    for (int g0 = 0; g0 < TP; g0 += 64) {
                        const int g = g0 + lane;
                        const int v = (g < TP && g <= cut) ? stepcnt[g] : 0;
                        int incl = v;
                        for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off); if (lane >= off) incl += u; }
                        if (g < TP) stepcnt[g] = carry + incl - v;
                        carry += __shfl(incl, 63);
                    }
                }
                const int SPt = (S + 63) >> 6, nwv = nthreads >> 6;
                int capk = prm.cap0;
                long long thrk = thr0;
                // A map that resized once or twice (the shipped profiles: 28 000 ids in a table that started at 32 768): ALL its
                // capacity stages in ONE walk over the tuples, a counter table per stage side by side in LDS (cap0 / 8 + 2 cap0 / 8 +
                // ... words) — a first occurrence of insertion rank r counts in every stage whose last rank is >= r.
                bool fused_stages = false;
                if (!single && SPt == 1) {
                    int nst = 0;
                    long long words = 0, thr_i = thr0;
                    int cap_i = prm.cap0;
                    while (nst < 3) {
                        words += cap_i >> 3;
                        nst++;
                        if (thr_i >= static_cast<long long>(n) - 1) break;
                        cap_i <<= 1; thr_i <<= 1;
                    }
                    if (thr_i >= static_cast<long long>(n) - 1 && words <= nib_words) {
                        fused_stages = true;
                        __syncthreads();
                        for (int i = tid; i < static_cast<int>(words); i += nthreads) nib[i] = 0u;
                        __syncthreads();
                        const long long e0 = (thr0 < static_cast<long long>(n) - 1) ? thr0 : static_cast<long long>(n) - 1;
                        const long long e1 = (2 * static_cast<long long>(thr0) < static_cast<long long>(n) - 1) ? 2 * static_cast<long long>(thr0) : static_cast<long long>(n) - 1;
                        const int o1 = prm.cap0 >> 3, o2 = o1 + (prm.cap0 >> 2);
                        bool hit = false;
                        constexpr int kTU = 4;
                        for (int ts0 = wave; ts0 < TP; ts0 += nwv * kTU) {      // wave-uniform: the ballots need every lane
                            if (ts0 > cut) break;
                            uint16_t vb[kTU];
                            int32_t ib[kTU];
#pragma unroll
                            for (int u = 0; u < kTU; u++) {
                                const int ts = ts0 + u * nwv;
                                const bool in = ts < TP && ts <= cut && lane < S;
                                vb[u] = in ? tscore[ts * S + lane] : static_cast<uint16_t>(0);
                                ib[u] = in ? tup[ts * S + lane] : -1;
                            }
#pragma unroll
                            for (int u = 0; u < kTU; u++) {
                                const int ts = ts0 + u * nwv;
                                if (ts >= TP || ts > cut) break;                 // wave-uniform
                                const bool f = (vb[u] & kFirstFlag) != 0;
                                const unsigned long long bm = __ballot(f);
                                const long long rank = stepcnt[ts] + __popcll(bm & ((1ull << lane) - 1ull));
                                if (!f) continue;
                                const uint32_t sp = spread_of(ib[u]);
                                if (rank <= e0) {
                                    const uint32_t b = sp & static_cast<uint32_t>(prm.cap0 - 1), sh = (b & 7u) * 4u;
                                    hit = hit || (((atomicAdd(&nib[b >> 3], 1u << sh) >> sh) & 15u) >= 8u);
                                }
                                if (nst >= 2 && rank <= e1) {
                                    const uint32_t b = sp & static_cast<uint32_t>(2 * prm.cap0 - 1), sh = (b & 7u) * 4u;
                                    hit = hit || (((atomicAdd(&nib[o1 + (b >> 3)], 1u << sh) >> sh) & 15u) >= 8u);
                                }
                                if (nst >= 3) {                                  // the last stage: every rank
                                    const uint32_t b = sp & static_cast<uint32_t>(4 * prm.cap0 - 1), sh = (b & 7u) * 4u;
                                    hit = hit || (((atomicAdd(&nib[o2 + (b >> 3)], 1u << sh) >> sh) & 15u) >= 8u);
                                }
                            }
                        }
                        if (hit) s_tree = 1;
                        __syncthreads();
                    }
                }
                for (int stage = 0; stage < 24 && !fused_stages; stage++) {
                    const long long endk = (thrk < static_cast<long long>(n) - 1) ? thrk : static_cast<long long>(n) - 1;   // last rank of this stage
                    __syncthreads();
                    for (int i = tid; i < (capk >> 3); i += nthreads) nib[i] = 0u;
                    __syncthreads();
                    bool hit = false;
                    if (single) {
                        constexpr int kTU = 4;
                        for (int j0 = tid; j0 < prm.max_tuples; j0 += nthreads * kTU) {
                            uint16_t vb[kTU];
                            int32_t ib[kTU];
#pragma unroll
                            for (int u = 0; u < kTU; u++) {
                                const int j = j0 + u * nthreads;
                                vb[u] = (j < prm.max_tuples) ? tscore[j] : static_cast<uint16_t>(0);
                                ib[u] = (j < prm.max_tuples) ? tup[j] : -1;
                            }
#pragma unroll
                            for (int u = 0; u < kTU; u++) {
                                if (!(vb[u] & kFirstFlag)) continue;
                                const uint32_t b = spread_of(ib[u]) & static_cast<uint32_t>(capk - 1);
                                const uint32_t sh = (b & 7u) * 4u;
                                hit = hit || (((atomicAdd(&nib[b >> 3], 1u << sh) >> sh) & 15u) >= 8u);
                            }
                        }
                    } else if (SPt == 1) {
                        // (one 64-id piece per probe step: the tuple words of four steps of a wave are requested together)
                        constexpr int kTU = 4;
                        for (int ts0 = wave; ts0 < TP; ts0 += nwv * kTU) {      // wave-uniform: the ballots need every lane
                            if (ts0 > cut) break;
                            uint16_t vb[kTU];
                            int32_t ib[kTU];
#pragma unroll
                            for (int u = 0; u < kTU; u++) {
                                const int ts = ts0 + u * nwv;
                                const bool in = ts < TP && ts <= cut && lane < S;
                                vb[u] = in ? tscore[ts * S + lane] : static_cast<uint16_t>(0);
                                ib[u] = in ? tup[ts * S + lane] : -1;
                            }
#pragma unroll
                            for (int u = 0; u < kTU; u++) {
                                const int ts = ts0 + u * nwv;
                                if (ts >= TP || ts > cut) break;                 // wave-uniform
                                const bool f = (vb[u] & kFirstFlag) != 0;
                                const unsigned long long bm = __ballot(f);
                                const int rank = stepcnt[ts] + __popcll(bm & ((1ull << lane) - 1ull));
                                if (!f || rank > endk) continue;
                                const uint32_t b = spread_of(ib[u]) & static_cast<uint32_t>(capk - 1);
                                const uint32_t sh = (b & 7u) * 4u;
                                hit = hit || (((atomicAdd(&nib[b >> 3], 1u << sh) >> sh) & 15u) >= 8u);
                            }
                        }
                    } else {
                        for (int ts = wave; ts < TP; ts += nwv) {          // wave-uniform: the ballots need every lane
                            if (ts > cut) break;
                            int carry = stepcnt[ts];
                            for (int pc = 0; pc < SPt; pc++) {
                                const int pos = pc * 64 + lane;
                                const int j = ts * S + pos;
                                const bool f = (pos < S) && (tscore[j] & kFirstFlag);
                                const unsigned long long bm = __ballot(f);
                                const int rank = carry + __popcll(bm & ((1ull << lane) - 1ull));
                                carry += __popcll(bm);
                                if (!f || rank > endk) continue;
                                const uint32_t b = spread_of(tup[j]) & static_cast<uint32_t>(capk - 1);
                                const uint32_t sh = (b & 7u) * 4u;
                                hit = hit || (((atomicAdd(&nib[b >> 3], 1u << sh) >> sh) & 15u) >= 8u);
                            }
                        }
                    }
                    if (hit) s_tree = 1;
                    __syncthreads();
                    if (s_tree || thrk >= static_cast<long long>(n) - 1) break;     // block-uniform
                    capk <<= 1;
                    thrk <<= 1;
                    if (capk > capf) break;                                        // (n <= thr of the final table: not reached)
                }
            }
            if (!use_nib) {
                uint4* h4 = reinterpret_cast<uint4*>(ht);
                for (int i = tid; i < prm.ht_size / 4; i += nthreads) h4[i] = make_uint4(0, 0, 0, 0);
            }
            __syncthreads();
            if (!use_nib) for (int j = tid; j < prm.max_tuples; j += nthreads) {
                if (!(tscore[j] & kFirstFlag)) continue;
                const uint32_t b0 = spread_of(tup[j]) & static_cast<uint32_t>(prm.cap0 - 1);
                if (atomicAdd(&ht[b0 & ht_mask], 1u) >= 8u) s_suspect = 1;
            }
            __syncthreads();
            const bool exact0 = single && prm.ht_size >= prm.cap0;   // no folding, one stage: the counters ARE the bins
            if (use_nib) { }
            else if (s_suspect && exact0) { if (tid == 0) s_tree = 1; }
            else if (s_suspect) {
                // exact pass.  Insertion rank of a first occurrence = distinct ids of earlier probe steps + earlier first
                // occurrences inside its own step (tuples of a step are laid out in insertion order).
                if (wave == 0) {     // exclusive prefix of stepcnt over the steps that ran
                    int carry = 0;
                    for (int g0 = 0; g0 < TP; g0 += 64) {
                        const int g = g0 + lane;
                        const int v = (g < TP && g <= cut) ? stepcnt[g] : 0;
                        int incl = v;
                        for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off); if (lane >= off) incl += u; }
                        if (g < TP) stepcnt[g] = carry + incl - v;
                        carry += __shfl(incl, 63);
                    }
                }
                const int SPt = (S + 63) >> 6, nwv = nthreads >> 6;
                int capk = prm.cap0;
                long long thrk = thr0;
                for (int stage = 0; stage < 24; stage++) {
                    const long long endk = (thrk < static_cast<long long>(n) - 1) ? thrk : static_cast<long long>(n) - 1;   // last rank of this stage
                    __syncthreads();
                    {
                        uint4* h4 = reinterpret_cast<uint4*>(ht);
                        const uint4 e4 = make_uint4(kHtEmpty, kHtEmpty, kHtEmpty, kHtEmpty);
                        for (int i = tid; i < prm.ht_size / 4; i += nthreads) h4[i] = e4;
                    }
                    __syncthreads();
                    for (int ts = wave; ts < TP; ts += nwv) {          // wave-uniform: the ballots need every lane
                        if (ts > cut) break;
                        int carry = stepcnt[ts];
                        for (int pc = 0; pc < SPt; pc++) {
                            const int pos = pc * 64 + lane;
                            const int j = ts * S + pos;
                            const bool f = (pos < S) && (tscore[j] & kFirstFlag);
                            const unsigned long long bm = __ballot(f);
                            const int rank = carry + __popcll(bm & ((1ull << lane) - 1ull));
                            carry += __popcll(bm);
                            if (!f || rank > endk) continue;
                            const uint32_t b = spread_of(tup[j]) & static_cast<uint32_t>(capk - 1);     // < 2^20
                            uint32_t slot = (b * 2654435761u) >> prm.ht_shift;
                            const uint32_t stp = ((b * 0x85EBCA6Bu) >> prm.ht_shift) | 1u;
                            for (int tries = 0; tries < prm.ht_size; tries++) {    // <= n <= 0.8 ht_size keys: an empty slot exists
                                uint32_t cur = ht[slot];
                                if (cur == kHtEmpty) {
                                    cur = atomicCAS(&ht[slot], kHtEmpty, (b << 11) | 1u);
                                    if (cur == kHtEmpty) break;                    // created with count 1
                                }
                                if ((cur >> 11) == b) {                            // the key of a slot never changes
                                    if ((cur & 2047u) >= 9u || (atomicAdd(&ht[slot], 1u) & 2047u) + 1u >= 9u) s_tree = 1;
                                    break;
                                }
                                slot = (slot + stp) & ht_mask;
                            }
                        }
                    }
                    __syncthreads();
                    if (s_tree || thrk >= static_cast<long long>(n) - 1) break;     // block-uniform
                    capk <<= 1;
                    thrk <<= 1;
                    if (capk > (1 << kBucketBits)) break;                          // beyond the bucket field (host rejects such caps)
                }
            }
            __syncthreads();
        }
        const bool treeified = (s_tree != 0);
        if (tid == 0) {
            if (treeified && prm.unmodelled) atomicAdd(prm.unmodelled, 1);
            prm.out_count[qi] = treeified ? kRouteUnmodelled : nout;
            if (prm.out_kept) prm.out_kept[qi] = n;
            if (prm.out_raw) prm.out_raw[qi] = n + s_raw;
        }
        // LDS-only barrier: the next query may reuse the scratch, but nothing has to wait for this query's global
        // result stores (a __syncthreads() would add their full write latency to every query)
        __syncthreads();
        FSP_STAMP(6);
    }
#undef FSP_STAMP
#undef FSP_TS
}

template <bool kLds, int kThreads>
__global__ __launch_bounds__(kThreads, 4) void route_select_kernel(RouteParams prm, const int4* __restrict__ probe_in,
                                                                const int32_t* __restrict__ nprobe_in) {
    extern __shared__ __align__(16) unsigned char smem[];
    // list mode (hand-back of the bounded select): normally the list is empty — leave before anything else is touched
    if (prm.qlist && *prm.qcount == 0) return;
    const int TP = prm.TD * prm.P;
    const int64_t nq_eff = prm.qlist ? static_cast<int64_t>(*prm.qcount) : prm.nq;
    for (int64_t qq = blockIdx.x; qq < nq_eff; qq += gridDim.x) {
        const int64_t qi = prm.qlist ? static_cast<int64_t>(prm.qlist[qq]) : qq;
        route_select_query<kLds, kThreads>(prm, smem, static_cast<int>(blockIdx.x), probe_in + qi * TP, nprobe_in + qi * prm.TD, qi);
    }
}

}  // namespace fspann
