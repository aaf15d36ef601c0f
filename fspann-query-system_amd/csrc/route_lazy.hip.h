// route_lazy.hip.h — Route select when only the first `limit` entries of the Java-ordered candidate list are
// wanted (QSI stage A.5, QSI:169-214) and neither counter (lastCandKept / rawSeen) is asked for.
//
// The list is ordered by (score, HashMap bucket, first insertion) and an id's score is the MINIMUM Hamming distance
// over the probed partitions that hold it (PIS:726-753), every id of one probed partition carrying that partition's
// distance.  So the first `limit` entries are decided by the probed partitions with the SMALLEST distances, and
// inside the distance level that crosses `limit` by the ids with the smallest buckets:
//
//   1. sort the <= T*D*P probed partitions of the query by distance (all-pairs rank, a few hundred elements);
//   2. walk the distance LEVELS in ascending order, keeping an LDS hash  id -> (min distance, bucket):
//        a. whole levels whose tuples cannot reach `limit` distinct ids are inserted as they are;
//        b. the level that can cross `limit` is pre-filtered: histogram of its not-yet-present ids over the top
//           10 bits of the bucket, cut at the `need`-th id, and only ids at or below the cut are inserted (the
//           histogram counts an id once per partition holding it, so if the cut yields too few DISTINCT ids it is
//           moved up and the step repeats; when the whole level is in, the walk goes on to the next level).
//      Afterwards every id that can be among the first `limit` is in the hash with its exact score, and the hash
//      holds about `limit` + (level size / 1024) entries — not T*D*P*S;
//   3. rank the entries by (score, bucket) — 32-bit keys, all-pairs, no barriers — and write ranks < limit;
//   4. "first insertion" matters only between entries that agree on (score, bucket) — a handful per query.  For
//      those the reference's insertion sequence is recomputed exactly from the inverse id map (inv[td][id] =
//      position of id in table td's id list): the first table, in Java order, whose probed partitions cover that
//      position.  That also accounts for occurrences in partitions this kernel never loaded.
//
// Preconditions (checked by the host, otherwise route_select_kernel runs): the HARD_CAP cannot trigger
// (T*D*P*S < HARD_CAP), the HashMap never resizes (T*D*P*S <= 0.75 * initial capacity, so the bucket of an id does
// not depend on how many ids were inserted), no table holds an id twice, out_kept == out_raw == NULL, limit <= 1024.
// A query whose entries do not fit (degenerate hashCodes: one bucket bin holding hundreds of ids) is appended to an
// overflow list and redone by route_select_kernel (qlist mode) — same results, just slower.
#pragma once
#include "route.hip.h"

namespace fspann {

constexpr uint64_t kLzEmpty = ~0ull;
constexpr int kLzThreads = 256;
// Three size classes (template parameter kEnt = distinct ids one query may hold before it is handed over): 512 for limit <= 256
// (19.6 KB of LDS, 6 workgroups per CU: the kernel's throughput follows its workgroups per CU), 1024 for
// limit <= 512 (33 KB, 4 per CU) and 2048 for limit <= 1024 (BASELINE config #4's B; 63 KB, 2 per CU).
constexpr int kLzEntriesMax = 1024;
constexpr int kLzCollMax = 256;     // entries sharing (score, bucket) with another one
constexpr int lz_ht_size(int kEnt) { return 2 * kEnt; }          // hash slots (64-bit entries): load <= 0.5
constexpr int lz_sort_max(int kEnt) { return kEnt - 128; }       // entries the rank pass takes (16 slices padded to multiples of 8 stay < kEnt)
constexpr size_t lz_lds_bytes(int kEnt, int TD, int P) {         // dynamic LDS of route_lazy_run (host and device agree through this)
    return static_cast<size_t>(lz_ht_size(kEnt)) * 8 + static_cast<size_t>(TD) * P * 16 + static_cast<size_t>(kEnt) * 4 + 4096 +
           (static_cast<size_t>(TD) * P + 2) * 8 + (static_cast<size_t>(TD) * P + 1) * 4 + 8 + static_cast<size_t>(TD) * 8 +
           ((static_cast<size_t>(TD) * P * 2 + 3) & ~size_t(3)) + static_cast<size_t>(kEnt) * 2 * 3 + static_cast<size_t>(kEnt) * 4 +
           static_cast<size_t>(TD) * P * 8 + static_cast<size_t>(TD) * 4 + 16 + static_cast<size_t>(TD) * 4 + static_cast<size_t>(TD) * P * 4;
}
// probed partitions in flight per wave (lz_stage_u) and partitions of the crossing level one wave keeps in registers (lz_keep).
// (Halving both fits the small class into 64 registers = 8 workgroups per CU, but a lone launch then takes 52 us instead of 42
// and three overlapped ones gain nothing: measured, not used.)
constexpr int lz_stage_u(int) { return 4; }
constexpr int lz_keep(int) { return 8; }

// wave_find_cut for exactly 1024 bins starting at a 16-byte aligned address: each lane takes 16 bins with four
// 16-byte reads and walks ITS bins out of registers (no second LDS pass).  One wave; result in every lane.
__device__ __forceinline__ int wave_cut1024(const int32_t* bins, int need, int lane, int* before) {
    const int4* b4 = reinterpret_cast<const int4*>(bins) + lane * 4;
    const int4 v0 = b4[0], v1 = b4[1], v2 = b4[2], v3 = b4[3];
    const int vals[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += vals[i];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    const int excl = incl - sum;
    const unsigned long long bm = __ballot((incl >= need) && (excl < need));
    if (bm == 0) {   // total < need: everything before the last bin
        *before = __shfl(incl, 63) - __shfl(v3.w, 63);
        return 1023;
    }
    const int src = __ffsll(static_cast<long long>(bm)) - 1;
    int cum = excl, bb = 15;
    bool found = false;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (!found) {
            if (cum + vals[i] >= need) { bb = i; found = true; }
            else cum += vals[i];
        }
    }
    *before = __shfl(cum, src);
    return __shfl(lane * 16 + bb, src);
}

// The bounded select of the queries q_first, q_first + q_stride, ... < prm.nq by ONE workgroup of kThreads threads
// (`smem` = its dynamic LDS, `block_id` = its slice of the global fallback arena).  Called by route_select_lazy_kernel
// and by the route role of tick_kernel (tick.hip.h).
// kChk: built with the exact treeify check (step 0; it runs when prm.bin16 is set).  A template parameter because the check's
// loads live across the ordering step: in the 512-entry class (80 registers at six workgroups per CU) they would spill, and any
// scratch use costs every dispatch of the stream — that class is built without it, and a checked Route takes the 1024-entry class.
// kTD / kP (0 = run-time values): the tables x probes shape as compile-time constants.  With them every LDS array base, trip count
// and division by T*D*P below is an immediate; left to run time the kernel computes ~60 scalars in its prologue, keeps them for the
// whole query and — beyond the 102 scalar registers it has — parks them in vector-register lanes (v_writelane / v_readlane).
// The host launches the specialised build for BASELINE config #2 / #3's shape (16 tables x 5 probes, blocks of 64).
template <int kThreads, int kEnt = kLzEntriesMax, bool kChk = true, int kTD = 0, int kP = 0>
__device__ __forceinline__ void route_lazy_run(const RouteParams& prm, unsigned char* smem, const int64_t q_first, const int64_t q_stride, const int block_id) {
    int4* probe_in = prm.probe_g;          // not __restrict__: a handed-over query's lists are written here and read back
    int32_t* nprobe_in = prm.nprobe_g;
    int tid = threadIdx.x;                      // not const: see the register note behind the probe
    constexpr int nthreads = kThreads;
    constexpr int nwv = kThreads / 64;
    int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: a scalar — what derives from it is scalar work
    const int TD = kTD > 0 ? kTD : prm.TD, P = kP > 0 ? kP : prm.P, S = (kTD > 0) ? 64 : prm.S;
    const int TP = TD * P;
    const int SP = (S + 63) >> 6;                 // 64-lane pieces per partition

    constexpr int kLzHtSize = lz_ht_size(kEnt), kLzEntries = kEnt, kLzSortMax = lz_sort_max(kEnt);
    constexpr int kLzStageU = lz_stage_u(kEnt), kLzKeep = lz_keep(kEnt);
    static_assert(kEnt == 512 || kEnt == 1024 || kEnt == 2048, "size classes");
    static_assert(!(kChk && kEnt == 512), "the 512-entry class has no registers for the check");
    size_t o = 0;
    uint64_t* ht = reinterpret_cast<uint64_t*>(smem + o);        o += static_cast<size_t>(kLzHtSize) * 8;
    int4* plist = reinterpret_cast<int4*>(smem + o);             o += static_cast<size_t>(TP) * 16;
    uint32_t* pre = reinterpret_cast<uint32_t*>(smem + o);       o += static_cast<size_t>(kEnt) * 4;   // (score << 20 | bucket) per entry
    int32_t* bins = reinterpret_cast<int32_t*>(smem + o);        o += 1024 * 4;
    uint2* pkv = reinterpret_cast<uint2*>(smem + o);             o += (static_cast<size_t>(TP) + 2) * 8;   // {distance << 16 | probe index, size}, padded to even
    int32_t* pcs = reinterpret_cast<int32_t*>(smem + o);         o += (static_cast<size_t>(TP) + 1) * 4;
    o = (o + 7) & ~size_t(7);
    int64_t* ids_base = reinterpret_cast<int64_t*>(smem + o);    o += static_cast<size_t>(TD) * 8;
    int64_t* pbase = reinterpret_cast<int64_t*>(smem + o);       o += static_cast<size_t>(TP) * 8;   // per probed partition: position of its first id in ids_bk
    uint16_t* ord = reinterpret_cast<uint16_t*>(smem + o);       o += (static_cast<size_t>(TP) * 2 + 3) & ~size_t(3);
    uint16_t* ulist = reinterpret_cast<uint16_t*>(smem + o);     o += static_cast<size_t>(kLzEntries) * 2;   // hash slots of the entries
    uint16_t* rk = reinterpret_cast<uint16_t*>(smem + o);        o += static_cast<size_t>(kEnt) * 2;   // entries per (score, bucket) rank: > 1 = collision
    uint16_t* lrank = reinterpret_cast<uint16_t*>(smem + o);     o += static_cast<size_t>(kEnt) * 2;   // (score, bucket) rank of each entry
    uint32_t* gk = reinterpret_cast<uint32_t*>(smem + o);        o += static_cast<size_t>(kEnt) * 4;   // the keys grouped by their top bits (rank pass)
    int32_t* nprobe_l = reinterpret_cast<int32_t*>(smem + o);    o += static_cast<size_t>(TD) * 4;   // [TD] fused probe: partitions probed per table
    int32_t* part0 = reinterpret_cast<int32_t*>(smem + o);       o += static_cast<size_t>(TD) * 4;   // first partition of each table (global numbering)
    int32_t* pglob = reinterpret_cast<int32_t*>(smem + o);       // [TP] global number of each probed partition: its row of bin16
    // collision records alias the histogram (free once the levels are in): element, prefix rank, id, sequence
    int32_t* c_elem = bins;
    int32_t* c_lt = bins + kLzCollMax;
    int32_t* c_id = bins + 2 * kLzCollMax;
    int32_t* c_seq = bins + 3 * kLzCollMax;

    __shared__ int s_u, s_R, s_ncoll, s_bad, s_b, s_cnt, s_short, s_susp;
    constexpr uint32_t ht_mask = kLzHtSize - 1;
    constexpr int ht_shift = (kEnt == 2048) ? 32 - 12 : (kEnt == 1024) ? 32 - 11 : 32 - 10;   // 32 - log2(hash slots)
    unsigned long long lt_mask = 0;

    for (int i = tid; i < TD; i += nthreads) { ids_base[i] = prm.tables[i].ids_base; part0[i] = static_cast<int32_t>(prm.tables[i].part_base); }
    for (int i = tid; i < kLzHtSize; i += nthreads) ht[i] = kLzEmpty;
    if (kChk && prm.bin16) for (int i = tid; i < 1024; i += nthreads) bins[i] = 0;   // the check's counters (step 0) start clear
    if (block_id == 0 && tid == 0) *prm.ovf_next = 0;   // the other counter, for the next call (stream-ordered after this one)
    __syncthreads();                                    // ids_base is read when the first query's probe list is completed

    // is `id` already an entry?  (an entry of an earlier level: its score is lower, this occurrence changes nothing)
    auto present = [&](int32_t id) -> bool {
        uint32_t slot = (static_cast<uint32_t>(id) * 2654435761u) >> ht_shift;
        const uint32_t stp = ((static_cast<uint32_t>(id) * 0x85EBCA6Bu) >> ht_shift) | 1u;
        for (int tries = 0; tries < kLzHtSize; tries++) {
            const uint64_t e = ht[slot];
            if (e == kLzEmpty) return false;
            if (static_cast<uint32_t>(e >> 32) == static_cast<uint32_t>(id)) return true;
            slot = (slot + stp) & ht_mask;
        }
        return false;
    };
    // ht[id] = min(ht[id], low) with low = score << 20 | bucket; returns true when the entry was created (*slot_out)
    auto insert = [&](int32_t id, uint32_t low, uint32_t* slot_out) -> bool {
        const uint64_t mine = (static_cast<uint64_t>(static_cast<uint32_t>(id)) << 32) | low;
        uint32_t slot = (static_cast<uint32_t>(id) * 2654435761u) >> ht_shift;
        const uint32_t stp = ((static_cast<uint32_t>(id) * 0x85EBCA6Bu) >> ht_shift) | 1u;
        // <= kLzEntries < kLzHtSize entries and an odd step over a power-of-two table: an empty slot or the id itself
        // is met within kLzHtSize trips (the bound only guards against a host-side slip)
        for (int tries = 0; tries < kLzHtSize; tries++) {
            const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&ht[slot]), kLzEmpty, mine);
            if (old == kLzEmpty) { *slot_out = slot; return true; }
            if (static_cast<uint32_t>(old >> 32) == static_cast<uint32_t>(id)) {
                if (static_cast<uint32_t>(old) > low) atomicMin(reinterpret_cast<unsigned long long*>(&ht[slot]), mine);
                return false;
            }
            slot = (slot + stp) & ht_mask;
        }
        s_bad = 1;
        return false;
    };

#ifdef FSPANN_DEBUG_STAMPS
#define LZ_STAMP(i) do { if (prm.dbg && tid == 0 && qi == block_id) prm.dbg[block_id * 16 + (i)] = wall_clock64(); } while (0)
#else
#define LZ_STAMP(i) do { } while (0)
#endif
// FSPANN_LZ_STOP_AFTER=n (tools/pmc_phases.sh only, never a release build): every query stops behind phase n (1 probe, 2 probe
// ordering, 3 level walk, 5 rank) — results are garbage, the instruction counters of the truncated kernel are what is wanted.
#ifdef FSPANN_LZ_STOP_AFTER   /* (a plain `if`, not do { } while (0): its `continue` is the query loop's) */
#define LZ_STOP(n) if (FSPANN_LZ_STOP_AFTER == (n) && prm.nq > 0) { __syncthreads(); if (tid == 0) prm.out_count[qi] = 0;               \
        const int used_ = min(s_u, kLzEntries); for (int i_ = tid; i_ < used_; i_ += nthreads) ht[ulist[i_]] = kLzEmpty; __syncthreads(); continue; }
#else
#define LZ_STOP(n)
#endif
// Every live tuple of the sorted probes [RA, RB): BODY sees `id` (>= 0 when live and not deleted, else -1), its bucket
// field `bf` and the partition's distance `sc`;
// kLzStageU partitions are in flight per wave (one coalesced 256-byte id row each).  Trip counts are wave-uniform.
#define LZ_FOR_TUPLES(RA, RB, BODY)                                                                                   \
    do {                                                                                                              \
        const int nit_ = ((RB) - (RA)) * SP;                                                                          \
        for (int it0_ = wave; it0_ < nit_; it0_ += nwv * kLzStageU) {                                                 \
            uint64_t entv_[kLzStageU];                                                                                \
            int scv_[kLzStageU];                                                                                      \
            _Pragma("unroll") for (int v_ = 0; v_ < kLzStageU; v_++) {                                                \
                const int it_ = it0_ + v_ * nwv;                                                                      \
                entv_[v_] = kLzEmpty; scv_[v_] = 0;                                                                   \
                if (it_ < nit_) {                                                                                     \
                    const int rr_ = (RA) + it_ / SP, pos_ = (it_ % SP) * 64 + lane;                                   \
                    const int pi_ = ord[rr_];                                                                         \
                    const int4 pr_ = plist[pi_];                                                                      \
                    scv_[v_] = pr_.y;                                                                                 \
                    if (pos_ < pr_.w) entv_[v_] = prm.ids_bk[pbase[pi_] + pos_];                                      \
                }                                                                                                     \
            }                                                                                                         \
            _Pragma("unroll") for (int v_ = 0; v_ < kLzStageU; v_++) {                                                \
                if (it0_ + v_ * nwv >= nit_) continue;                                                                \
                int32_t id = static_cast<int32_t>(entv_[v_] >> 32);   /* -1 when no tuple */                           \
                const uint32_t bf = static_cast<uint32_t>(entv_[v_]); (void)bf;                                       \
                const int sc = scv_[v_]; (void)sc;                                                                    \
                if (id >= 0 && prm.deleted_bits && ((prm.deleted_bits[id >> 5] >> (id & 31)) & 1u)) id = -1; /* PIS:739 */ \
                BODY                                                                                                  \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
// insert + record the new entries of this wave-trip (one LDS atomic per wave)
#define LZ_INSERT(COND, LOW)                                                                                          \
    do {                                                                                                              \
        bool created_ = false;                                                                                        \
        uint32_t slot_ = 0;                                                                                           \
        if (COND) created_ = insert(id, (LOW), &slot_);                                                               \
        const unsigned long long bm_ = __ballot(created_);                                                            \
        if (bm_) {                                                                                                    \
            int base_ = 0;                                                                                            \
            if (lane == 0) base_ = atomicAdd(&s_u, __popcll(bm_));                                                    \
            base_ = __shfl(base_, 0);                                                                                 \
            const int at_ = base_ + __popcll(bm_ & lt_mask);                                                          \
            if (created_ && at_ < kLzEntries) ulist[at_] = static_cast<uint16_t>(slot_);                              \
        }                                                                                                             \
    } while (0)

    for (int64_t qi = q_first; qi < prm.nq; qi += q_stride) {
        LZ_STAMP(0);
        // ---- probe list of this query; unused steps get an impossible partition and sort last ------------------
        if (prm.probe_G > 0) {
            // fused probe (route_probe_table): one group of G lanes per table, all tables of the query side by side
            const int G = (kTD > 0) ? 16 : prm.probe_G, lgG = 31 - __clz(G), ngroups = nthreads >> lgG;      // G: a power of two (16 whenever the probe is fused)
            const int grp_in_wave = lane >> lgG, gl = lane & (G - 1), grp = tid >> lgG;
            int32_t* w3 = reinterpret_cast<int32_t*>(pre) + grp * (2 * P - 1) * 3;     // `pre` is free until the rank pass
            for (int t0 = 0; t0 < TD; t0 += ngroups) {                                 // block-uniform trip count
                const int td = t0 + grp;
                const bool act = td < TD;
                const int tdc = act ? td : 0;
                const int Wc = (kTD > 0) ? 1 : prm.W;
#ifdef FSPANN_DEBUG_STAMPS
                // FSPANN_PROBE_STAMPS build (tools/route_probe_stamps.py): the first lane group of the first query of a workgroup stamps the
                // probe's rounds into the second half of the workgroup's debug row pair (dbg rows are 16 words; rows grid..2*grid-1)
                long long* pst = (prm.dbg && tid == 0 && qi == block_id && t0 == 0) ? prm.dbg + (static_cast<long long>(gridDim.x) + block_id) * 16 : nullptr;
#else
                long long* pst = nullptr;
#endif
                const int np = route_probe_table<int4*, kP, (kTD > 0 ? 1 : 0)>(prm, act, prm.codes + (qi * TD + tdc) * Wc, prm.tables[tdc], G, gl, grp_in_wave,
                                                                             w3, plist + tdc * P, pst);
                if (act && gl == 0) nprobe_l[td] = np;
            }
            __syncthreads();
            for (int i = tid; i < TP; i += nthreads) {
                const int td = i / P, step = i - td * P;
                int4 e = make_int4(-1, 0x7FFF, 0, 0);
                if (step < nprobe_l[td]) e = plist[i];
                plist[i] = e;
                pbase[i] = ids_base[td] + e.z;      // (a division by P per tuple trip otherwise: the table of a probe is i / P)
                pglob[i] = part0[td] + e.x;
                pkv[i] = make_uint2((static_cast<uint32_t>(e.y) << 16) | static_cast<uint32_t>(i), static_cast<uint32_t>(e.w));
            }
        } else {
            for (int i = tid; i < TP; i += nthreads) {
                const int td = i / P, step = i - td * P;
                int4 e = make_int4(-1, 0x7FFF, 0, 0);
                if (step < nprobe_in[qi * TD + td]) e = probe_in[qi * TP + i];
                plist[i] = e;
                pbase[i] = ids_base[td] + e.z;
                pglob[i] = part0[td] + e.x;
                pkv[i] = make_uint2((static_cast<uint32_t>(e.y) << 16) | static_cast<uint32_t>(i), static_cast<uint32_t>(e.w));
            }
        }
        if (tid == 0) { s_u = 0; s_R = TP; s_ncoll = 0; s_bad = 0; s_short = 0; s_susp = 0; pkv[TP] = make_uint2(0xFFFFFFFFu, 0u); }
        __syncthreads();
        LZ_STAMP(1);
        LZ_STOP(1)
        // Register note: everything below indexes by tid / lane / wave, and the compiler would compute all of those per-thread
        // offsets ONCE before the query loop and keep them alive through the probe above (its peak: a window of key ranges,
        // codes and id ranges per lane) — enough to spill.  Redefining the three here (the asm changes nothing) ties every
        // derived value to this point of the iteration.
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        lt_mask = (1ull << lane) - 1ull;
        // ---- 0. exact treeify check, first half (prm.bin16 != null: opaque ids, or asked for) ----------------------------
        // bestScore treeifies a bin when NINE distinct live ids of the probed partitions share it (HashMap.putVal at cap0, no
        // resize in reach: PIS:619) — including ids of partitions the walk below never loads.  The bins of ALL ids of the probed
        // partitions are read here (bin16: one 128-byte row per 64-id partition, requested before the ordering below and counted
        // after it) into 4-bit counters over the bins folded to 13 bits (the histogram's 4 KB, clear at this point): an id is
        // counted once per table holding it and several bins share a counter, so a counter below nine PROVES its bins stay
        // chains.  A counter that reaches nine is looked at exactly (second half, behind the barrier): rare.
        constexpr int kChkU = 5;
        uint2 chk_v[kChkU];
        int chk_n[kChkU];
        const int chk_cps = prm.bin16_shift - 2;                       // 8-byte chunks (four bins) per partition row, log2
        const int chk_items = (kChk && prm.bin16) ? (TP << chk_cps) : 0;
        const bool chk_once = chk_items <= nthreads * kChkU;           // everything in one round of loads (BASELINE shapes)
        auto chk_load = [&](const int it0) {
#pragma unroll
            for (int v = 0; v < kChkU; v++) {
                const int it = it0 + v * nthreads;
                chk_v[v] = make_uint2(0u, 0u); chk_n[v] = 0;
                if (it < chk_items) {
                    const int pi = it >> chk_cps, ch = it & ((1 << chk_cps) - 1);
                    const int4 pr = plist[pi];
                    const int rem = pr.w - ch * 4;
                    if (pr.x >= 0 && rem > 0) {
                        chk_v[v] = *reinterpret_cast<const uint2*>(prm.bin16 + (static_cast<int64_t>(pglob[pi]) << prm.bin16_shift) + ch * 4);
                        chk_n[v] = min(rem, 4);
                    }
                }
            }
        };
        auto chk_count = [&]() {
            bool sus = false;
#pragma unroll
            for (int v = 0; v < kChkU; v++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k < chk_n[v]) {
                        const uint32_t b16 = ((k < 2 ? chk_v[v].x : chk_v[v].y) >> ((k & 1) * 16)) & 0xFFFFu;
                        const uint32_t sh = (b16 & 7u) * 4u;
                        const uint32_t old = atomicAdd(reinterpret_cast<uint32_t*>(bins) + ((b16 & 8191u) >> 3), 1u << sh);
                        sus = sus || (((old >> sh) & 15u) >= 8u);    // this id is the ninth (or later) of its counter
                    }
                }
            }
            if (sus) s_susp = 1;
        };
        if (chk_items > 0 && chk_once) chk_load(tid);
        // ---- 1. order the probed partitions by (distance, Java order); prefix sums of their sizes ----------------
        for (int i = tid; i < TP; i += nthreads) {
            const uint2 me = pkv[i];
            const uint32_t mk = me.x;
            int rank = 0, cum = 0;
            const uint4* kv4 = reinterpret_cast<const uint4*>(pkv);    // two probes per 16-byte LDS read (broadcast)
#pragma unroll 4
            for (int j = 0; j < (TP + 1) / 2; j++) {
                const uint4 kv = kv4[j];
                const bool b0 = kv.x < mk, b1 = kv.z < mk;
                rank += b0 + b1;
                cum += (b0 ? kv.y : 0u) + (b1 ? kv.w : 0u);
            }
            ord[rank] = static_cast<uint16_t>(i);
            pcs[rank] = cum;
            if ((mk >> 16) == 0x7FFFu) atomicMin(&s_R, rank);   // unused steps sort last: the first of them ends the list
            if (rank == TP - 1) pcs[TP] = cum + static_cast<int>(me.y);
        }
        if (chk_items > 0) {
            if (chk_once) chk_count();
            else for (int it0 = tid; it0 < chk_items; it0 += nthreads * kChkU) { chk_load(it0); chk_count(); }   // (block-uniform trips)
        }
        __syncthreads();
        LZ_STAMP(2);
        LZ_STOP(2)
        const int R = s_R;                           // valid probes occupy sorted positions [0, R)
        bool overflow = false;
        if (chk_items > 0 && s_susp) {
            // ---- 0. second half (rare): the ids behind the counters that reached nine, looked at exactly ----------------
            // Sweep A counts again into the cleared table and marks every counter whose increment is the ninth or later in a
            // bitmap (a counter passes nine before it can wrap, and a wrap only carries into its neighbour: more marks, never
            // fewer); sweep B collects every live entry of a marked counter as (true bin, id); then first occurrences of an id
            // (it comes once per table holding it) and, per TRUE bin, the distinct ids: nine or more = the JVM treeifies.
            uint32_t* mark = reinterpret_cast<uint32_t*>(ht);                  // 8192 bits = 128 slots (ht is all-empty here: restored below)
            uint64_t* cl = ht + 128;                                           // up to kChkList collected (bin << 32 | id)
            uint64_t* cf = cl + 256;                                           // first occurrences
            constexpr int kChkList = 256;
            static_assert(128 + 2 * 256 <= lz_ht_size(512), "the check's scratch fits the smallest hash table");
            for (int i = tid; i < 256; i += nthreads) mark[i] = 0u;
            for (int i = tid; i < 1024; i += nthreads) bins[i] = 0;
            __syncthreads();
            for (int it = tid; it < chk_items; it += nthreads) {
                const int pi = it >> chk_cps, ch = it & ((1 << chk_cps) - 1);
                const int4 pr = plist[pi];
                const int rem = pr.w - ch * 4;
                if (pr.x < 0 || rem <= 0) continue;
                const uint2 vv = *reinterpret_cast<const uint2*>(prm.bin16 + (static_cast<int64_t>(pglob[pi]) << prm.bin16_shift) + ch * 4);
                for (int k = 0; k < min(rem, 4); k++) {
                    const uint32_t b16 = ((k < 2 ? vv.x : vv.y) >> ((k & 1) * 16)) & 0xFFFFu, f = b16 & 8191u;
                    const uint32_t sh = (f & 7u) * 4u;
                    const uint32_t old = atomicAdd(reinterpret_cast<uint32_t*>(bins) + (f >> 3), 1u << sh);
                    if (((old >> sh) & 15u) >= 8u) atomicOr(&mark[f >> 5], 1u << (f & 31u));
                }
            }
            __syncthreads();
            for (int it = tid; it < chk_items; it += nthreads) {
                const int pi = it >> chk_cps, ch = it & ((1 << chk_cps) - 1);
                const int4 pr = plist[pi];
                const int rem = pr.w - ch * 4;
                if (pr.x < 0 || rem <= 0) continue;
                const uint2 vv = *reinterpret_cast<const uint2*>(prm.bin16 + (static_cast<int64_t>(pglob[pi]) << prm.bin16_shift) + ch * 4);
                for (int k = 0; k < min(rem, 4); k++) {
                    const uint32_t b16 = ((k < 2 ? vv.x : vv.y) >> ((k & 1) * 16)) & 0xFFFFu, f = b16 & 8191u;
                    if (!((mark[f >> 5] >> (f & 31u)) & 1u)) continue;
                    const int32_t id = prm.ids[pbase[pi] + ch * 4 + k];      // bin16 rows are in the id list's order
                    if (prm.deleted_bits && ((prm.deleted_bits[id >> 5] >> (id & 31)) & 1u)) continue;                 // PIS:739: never put
                    const int c = atomicAdd(&s_ncoll, 1);
                    if (c < kChkList) cl[c] = (static_cast<uint64_t>(b16) << 32) | static_cast<uint32_t>(id);
                }
            }
            __syncthreads();
            const int ncl = s_ncoll;
            if (ncl > kChkList) {
                overflow = true;                      // too many suspects to settle here: the full select checks exactly
            } else {
                for (int i = tid; i < ncl; i += nthreads) {
                    const uint64_t me = cl[i];
                    bool first = true;
                    for (int j = 0; j < i; j++) first = first && (cl[j] != me);      // the same id (hence the same bin) from another table
                    cf[i] = first ? me : ~0ull;
                }
                __syncthreads();
                for (int i = tid; i < ncl; i += nthreads) {
                    const uint64_t me = cf[i];
                    if (me == ~0ull) continue;
                    int same = 0;
                    for (int j = 0; j < ncl; j++) same += (cf[j] != ~0ull) && ((cf[j] >> 32) == (me >> 32));
                    if (same >= 9) s_bad = 1;         // nine distinct live ids in one bin: the JVM treeifies it
                }
            }
            __syncthreads();
            if (s_bad) overflow = true;
            for (int i = tid; i < 128 + 2 * kChkList; i += nthreads) ht[i] = kLzEmpty;
            if (tid == 0) s_ncoll = 0;
            __syncthreads();
        }
        int r0 = 0, u = 0;
        int dbg_outer = 0, dbg_inner = 0, dbg_reload = 0, dbg_nitb = 0;
        (void)dbg_outer; (void)dbg_inner; (void)dbg_reload; (void)dbg_nitb;   // read only in FSPANN_DEBUG_STAMPS builds
        bool force_full = false;
        // ---- 2. walk the distance levels ---------------------------------------------------------------------------
        while (r0 < R && u < prm.limit && !overflow) {
            const int need = prm.limit - u;
            // r_fit = last level boundary whose cumulated tuples stay <= need; r_nofit = the boundary after it (the end of
            // the level that can cross `limit`).  Every wave computes the same values: the loop is wave-uniform.
            int r_fit = r0, r_nofit = R;
            for (int c0 = r0 + 1; c0 <= R; c0 += 64) {
                const int r = c0 + lane;
                const bool bnd = (r <= R) && (r == R || (pkv[ord[r]].x >> 16) != (pkv[ord[r - 1]].x >> 16));
                const unsigned long long bb = __ballot(bnd);
                const unsigned long long bf = __ballot(bnd && (pcs[r] - pcs[r0] <= need));
                if (bf) r_fit = c0 + 63 - __clzll(static_cast<long long>(bf));
                if (bb & ~bf) { r_nofit = c0 + __ffsll(static_cast<long long>(bb & ~bf)) - 1; break; }   // later ones do not fit either
            }
            const int rb0 = r_fit, rb1 = (r_fit < R) ? r_nofit : R;   // [rb0, rb1): the level that can cross (may be empty)
            // one global round trip for both: the crossing level is requested now and stays in registers.  Its partitions
            // are bucket-sorted (ids_bk), and only about need/tuples of the level can be selected, so a PREFIX of L entries
            // per partition is read: twice the expected share + 8 (checked after the cut: a partition whose L-th entry
            // is still at or below the cut was read short -> the level is redone with whole partitions).
            const int npB = rb1 - rb0;
            // How many TUPLES it takes to find one new id, in sixteenths (16 = every tuple is a new id: iid data; tables that agree —
            // clustered data — offer the same id again and again, inside a level too): from the levels walked so far, later from
            // the crossing level's own last step.  The histogram below counts tuples, so a cut sized for `need` tuples yields
            // need / boost ids and the level took five steps on average (fourteen at worst) on clustered data — 21 us of a 47 us
            // query; sizing the cut (and the prefix read per partition) for need * boost tuples takes one or two.  Overshooting
            // is harmless: the cut always takes whole bins in order, and the ranking keeps the first `limit` entries.
            // (a ladder of compares, no division: 16, 24, 32, 48, 64, 128 sixteenths)
            auto tuples_per_id = [](const int tuples, const int ids) -> int {
                return (tuples >= 8 * ids) ? 128 : (tuples >= 4 * ids) ? 64 : (tuples >= 3 * ids) ? 48 : (tuples >= 2 * ids) ? 32 : (2 * tuples >= 3 * ids) ? 24 : 16;
            };
            int boost = 16;
            if (r0 > 0 && u > 0) boost = min(64, tuples_per_id(pcs[r0], u));
            int L = 64;
            if (npB > 0 && SP == 1 && !force_full) {
                const int tuplesB = max(pcs[rb1] - pcs[rb0], 1);
                const int want = (2 * 64 * ((need * boost) >> 4)) / tuplesB + 8;
                L = (want <= 8) ? 8 : (want <= 16) ? 16 : (want <= 32) ? 32 : 64;
            }
            const int Lsh = 31 - __clz(L);                      // log2(L)
            const int G = 64 >> Lsh;                            // partitions per wave trip
            const int nitB = (L == 64) ? npB * SP : (npB + G - 1) / G;
            const bool keep = (nitB > 0) && (nitB <= nwv * kLzKeep);
            dbg_outer++; dbg_reload += (nitB > 0 && !keep); dbg_nitb = max(dbg_nitb, nitB);
            int32_t idr[kLzKeep];
            uint32_t bfr[kLzKeep];   // bucket field; bit 31: last entry of a prefix that does not cover its partition
            if (keep) {
#pragma unroll
                for (int k = 0; k < kLzKeep; k++) {
                    const int it = wave + k * nwv;
                    idr[k] = -1; bfr[k] = 0;
                    if (it < nitB) {
                        int pidx, pos;
                        if (L == 64) { pidx = it / SP; pos = (it % SP) * 64 + lane; }
                        else { pidx = it * G + (lane >> Lsh); pos = lane & (L - 1); }
                        if (pidx < npB) {
                            const int pi = ord[rb0 + pidx];
                            const int4 pr = plist[pi];
                            if (pos < pr.w) {
                                const uint64_t ent = prm.ids_bk[pbase[pi] + pos];
                                idr[k] = static_cast<int32_t>(ent >> 32);
                                bfr[k] = static_cast<uint32_t>(ent) | ((L < 64 && pos == L - 1 && pr.w > L) ? 0x80000000u : 0u);
                            }
                        }
                    }
                }
            }
            for (int i = tid; i < 1024; i += nthreads) bins[i] = 0;
            if (r_fit > r0) {
                // a. whole levels, at most `need` tuples: u stays <= limit
                LZ_FOR_TUPLES(r0, r_fit, {
                    LZ_INSERT(id >= 0, (static_cast<uint32_t>(sc) << kBucketBits) | bf);
                });
            }
            __syncthreads();
            LZ_STAMP(8);
            u = s_u;
            if (s_bad) { overflow = true; break; }
            if (u >= prm.limit || rb0 >= R) {
                r0 = rb0;
                __syncthreads();                     // s_u has been read by everyone before it moves again
                continue;
            }
            // b. the level [rb0, rb1) can cross `limit`: pre-filter by bucket
            const uint32_t sc_hi = (pkv[ord[rb0]].x >> 16) << kBucketBits;     // the level's distance
            if (keep) {
#pragma unroll
                for (int k = 0; k < kLzKeep; k++) {
                    int32_t id = idr[k];
                    if (id >= 0 && prm.deleted_bits && ((prm.deleted_bits[id >> 5] >> (id & 31)) & 1u)) id = -1;   // PIS:739
                    if (id >= 0) {
                        if (present(id)) id = -1;                              // entry of an earlier level: nothing to do
                        else atomicAdd(&bins[(bfr[k] & 0xFFFFFu) >> (kBucketBits - 10)], 1);
                    }
                    idr[k] = id;
                }
            } else {
                LZ_FOR_TUPLES(rb0, rb1, {
                    if (id >= 0 && !present(id)) atomicAdd(&bins[bf >> (kBucketBits - 10)], 1);
                });
            }
            __syncthreads();
            LZ_STAMP(10);
            int lo = 0;
            bool redo_full = false;
            if (rb0 > 0 && u > 0) boost = min(64, tuples_per_id(pcs[rb0], u));     // (the whole levels just walked included)
            while (true) {
                // tuples asked of this step: the ids still missing times the boost, but never so many that the entries could
                // outgrow the class if every one of them were new (three quarters of the room; at least the missing ids)
                const int missing = prm.limit - u;
                const int ask = max(missing, min((missing * boost) >> 4, ((prm.lazy_cap - u) * 3) >> 2));
                if (wave == 0) {
                    int before = 0;
                    const int b = (lo == 0) ? wave_cut1024(bins, ask, lane, &before)
                                            : lo + wave_find_cut(bins + lo, 1024 - lo, ask, lane, &before);
                    if (lane == 0) { s_b = b; s_cnt = before + bins[b]; }
                }
                __syncthreads();
                LZ_STAMP(11);
                dbg_inner++;
                const int b = s_b;
                const int offered = s_cnt;                                   // tuples in the bins [lo, b]
                if (u + offered > prm.lazy_cap) { overflow = true; break; }   // lazy_cap <= kLzEntries
                if (keep && L < 64) {
                    bool shortp = false;
#pragma unroll
                    for (int k = 0; k < kLzKeep; k++)
                        shortp = shortp || ((bfr[k] & 0x80000000u) && static_cast<int>((bfr[k] & 0xFFFFFu) >> (kBucketBits - 10)) <= b);
                    if (__any(shortp) && lane == 0) s_short = 1;
                    __syncthreads();
                    if (s_short) { redo_full = true; break; }
                }
                if (keep) {
                    // all inserts of this wave first, then ONE LDS atomic for the wave's new entries.  An id in range is
                    // finished either way (created, or already an entry): idr = -2 marks "created", bfr then holds the slot
                    int total = 0;
#pragma unroll
                    for (int k = 0; k < kLzKeep; k++) {
                        const int bin = static_cast<int>((bfr[k] & 0xFFFFFu) >> (kBucketBits - 10));
                        if (idr[k] >= 0 && bin >= lo && bin <= b) {
                            uint32_t slot = 0;
                            const bool c = insert(idr[k], sc_hi | (bfr[k] & 0xFFFFFu), &slot);
                            idr[k] = c ? -2 : -1;
                            if (c) bfr[k] = slot;
                        }
                        total += __popcll(__ballot(idr[k] == -2));
                    }
                    if (total) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&s_u, total);
                        base = __shfl(base, 0);
#pragma unroll
                        for (int k = 0; k < kLzKeep; k++) {
                            const bool c = (idr[k] == -2);
                            const unsigned long long bm = __ballot(c);
                            const int at = base + __popcll(bm & lt_mask);
                            if (c && at < kLzEntries) ulist[at] = static_cast<uint16_t>(bfr[k]);
                            if (c) idr[k] = -1;
                            base += __popcll(bm);
                        }
                    }
                } else {
                    LZ_FOR_TUPLES(rb0, rb1, {
                        const int bin = static_cast<int>(bf >> (kBucketBits - 10));
                        LZ_INSERT(id >= 0 && bin >= lo && bin <= b, sc_hi | bf);
                    });
                }
                __syncthreads();
                LZ_STAMP(12);
                const int u_was = u;
                u = s_u;
                if (s_bad) { overflow = true; break; }
                if (u >= prm.limit || b >= 1023) break;
                boost = tuples_per_id(offered, max(u - u_was, 1));
                lo = b + 1;
                __syncthreads();                     // s_u / s_b have been read by everyone before they move again
            }
            if (overflow) break;
            if (redo_full) {                          // nothing of this level was inserted yet: same level again, whole partitions
                r0 = rb0;
                force_full = true;
                __syncthreads();
                if (tid == 0) s_short = 0;
                __syncthreads();
                continue;
            }
            force_full = false;
            r0 = rb1;
            __syncthreads();
        }
        LZ_STAMP(3);
        LZ_STOP(3)
        // ---- 3. rank the entries by (score, bucket) ---------------------------------------------------------------
        const int nsel = u;
        if (nsel > kLzSortMax) overflow = true;
        if (!overflow) {
            const int nout = min(nsel, prm.limit);
            // Rank of an entry = #{keys < its key} (equal keys share a rank and are settled in step 4).  All-pairs over the
            // ~270 keys was a quarter of this kernel's vector instructions — and with several launches in flight the kernel is
            // bound by vector issue, not latency.  Instead: counting sort on the keys' TOP bits — (score above
            // the first level's, six bucket bits) -> 1 024 groups of ~1 key —, then every entry counts the smaller keys of its OWN
            // group only.  Three more barriers, a tenth of the instructions.
            const uint32_t smin = pkv[ord[0]].x >> 16;                       // the lowest distance level: no entry scores below it
            auto coarse = [&](const uint32_t key) -> uint32_t { return min((key - (smin << kBucketBits)) >> (kBucketBits - 6), 1023u); };
            for (int i = tid; i < kEnt; i += nthreads) pre[i] = (i < nsel) ? static_cast<uint32_t>(ht[ulist[i]]) : 0xFFFFFFFFu;
            for (int i = tid; i < kEnt / 2; i += nthreads) reinterpret_cast<uint32_t*>(rk)[i] = 0u;
            for (int i = tid; i < 1024; i += nthreads) bins[i] = 0;
            __syncthreads();
            LZ_STAMP(4);
            for (int i = tid; i < nsel; i += nthreads) atomicAdd(&bins[coarse(pre[i])], 1);
            __syncthreads();
            if (wave == 0) {                        // exclusive prefix of the 1 024 counts, in place: 16 bins per lane
                int4* b4 = reinterpret_cast<int4*>(bins) + lane * 4;
                int4 v[4] = {b4[0], b4[1], b4[2], b4[3]};
                int* vals = reinterpret_cast<int*>(v);
                int sum = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) { const int t = vals[i]; vals[i] = sum; sum += t; }
                int incl = sum;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(incl, off);
                    if (lane >= off) incl += t;
                }
                const int excl = incl - sum;
#pragma unroll
                for (int i = 0; i < 16; i++) vals[i] += excl;
                b4[0] = v[0]; b4[1] = v[1]; b4[2] = v[2]; b4[3] = v[3];
            }
            __syncthreads();
            for (int i = tid; i < nsel; i += nthreads) {
                const uint32_t my = pre[i];
                gk[atomicAdd(&bins[coarse(my)], 1)] = my;          // afterwards bins[c] = END of group c = start of group c + 1
            }
            __syncthreads();
            for (int i = tid; i < nsel; i += nthreads) {
                const uint32_t my = pre[i];
                const uint32_t c = coarse(my);
                const int g1 = bins[c], g0 = c ? bins[c - 1] : 0;
                int lt = g0;
                for (int j = g0; j < g1; j++) lt += gk[j] < my;
                // equal keys share their rank: count the entries per rank (2 x u16 per word)
                atomicAdd(reinterpret_cast<uint32_t*>(rk) + (lt >> 1), 1u << ((lt & 1) * 16));
                lrank[i] = static_cast<uint16_t>(lt);
            }
            __syncthreads();
            // classify only: the global result stores come last (a workgroup barrier waits for outstanding stores, so a
            // store followed by a barrier costs a full HBM write latency in the middle of the query)
            for (int i = tid; i < nsel; i += nthreads) {
                const int lt = lrank[i];
                if (rk[lt] != 1) {   // shares (score, bucket) with another entry: settled below by insertion sequence
                    // nine ids in one HashMap bin: the JVM treeifies it and its iteration order is no longer insertion
                    // order.  The full select detects that exactly (route.hip.h, phase T): hand the query over.
                    if (rk[lt] >= 9) s_bad = 1;
                    const int c = atomicAdd(&s_ncoll, 1);
                    if (c < kLzCollMax) { c_elem[c] = i; c_lt[c] = lt; c_id[c] = static_cast<int32_t>(ht[ulist[i]] >> 32); c_seq[c] = 0x7FFFFFFF; }
                }
            }
            __syncthreads();
            LZ_STAMP(5);
            LZ_STOP(5)
            const int ncoll = s_ncoll;
#ifdef FSPANN_DEBUG_STAMPS
            if (tid == 0 && prm.dbg && qi == block_id) { prm.dbg[block_id * 16 + 13] = ncoll; prm.dbg[block_id * 16 + 15] = nsel; prm.dbg[block_id * 16 + 14] = dbg_outer | (dbg_inner << 8) | (dbg_reload << 16) | (static_cast<long long>(dbg_nitb) << 24); }
#endif
            if (ncoll > kLzCollMax || s_bad) {
                overflow = true;
            } else {
                if (ncoll > 0) {
                    // ---- 4. reference insertion sequence of the colliding entries: one (entry, table) pair per thread, the
                    // inverse-map loads of up to four pairs in flight together ----
                    const int npair = ncoll * TD;
                    for (int t0 = tid; t0 < npair; t0 += nthreads * 4) {
                        int32_t idxv[4];
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            const int t = t0 + v * nthreads;
                            idxv[v] = (t < npair) ? prm.inv[static_cast<int64_t>(t % TD) * prm.n_ids + c_id[t / TD]] : -1;
                        }
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            const int t = t0 + v * nthreads;
                            const int32_t idx = idxv[v];
                            if (idx < 0) continue;
                            const int c = t / TD, td = t - c * TD;
                            for (int step = 0; step < P; step++) {
                                const int4 e = plist[td * P + step];
                                if (e.x >= 0 && idx >= e.z && idx < e.z + e.w) { atomicMin(&c_seq[c], (td * P + step) * S + (idx - e.z)); break; }
                            }
                        }
                    }
                    for (int c = tid; c < ncoll; c += nthreads) c_elem[c] = static_cast<int32_t>(pre[c_elem[c]]);   // element -> its key
                    __syncthreads();
                    for (int c = tid; c < ncoll; c += nthreads) {
                        const int32_t my = c_elem[c];
                        const int myseq = c_seq[c];
                        int rank = c_lt[c];
#pragma unroll 4
                        for (int c2 = 0; c2 < ncoll; c2++) rank += (c_elem[c2] == my) && (c_seq[c2] < myseq);
                        if (rank < nout) {
                            prm.out_ids[qi * prm.out_cap + rank] = c_id[c];
                            if (prm.out_score) prm.out_score[qi * prm.out_cap + rank] = static_cast<int32_t>(static_cast<uint32_t>(my) >> kBucketBits);
                        }
                    }
                }
                // ---- results of the entries with a key of their own ----
                for (int i = tid; i < nsel; i += nthreads) {
                    const int lt = lrank[i];
                    if (lt < nout && rk[lt] == 1) {
                        const uint64_t e = ht[ulist[i]];
                        prm.out_ids[qi * prm.out_cap + lt] = static_cast<int32_t>(e >> 32);
                        if (prm.out_score) prm.out_score[qi * prm.out_cap + lt] = static_cast<int32_t>(static_cast<uint32_t>(e) >> kBucketBits);
                    }
                }
            }
            if (!overflow && tid == 0) prm.out_count[qi] = nout;
        }
        if (overflow) {
            // This query does not fit the bounded select (or one of its (score, bin) groups would be a treeified bin): it is
            // handed to the full select — route_select_kernel in list mode right behind this kernel (fspann_route_dev), or the
            // workgroup that refines this query in the next tick (tick.hip.h).  Until then its count says PENDING.
            // (Running the full select right here, inlined or as a call, costs this kernel scratch memory: measured 2x slower.)
            if (prm.probe_G > 0) {   // the full select reads the probe lists from global memory: hand this query's over
                for (int i = tid; i < TP; i += nthreads) probe_in[qi * TP + i] = plist[i];
                for (int i = tid; i < TD; i += nthreads) nprobe_in[qi * TD + i] = nprobe_l[i];
            }
            if (tid == 0) {
                prm.out_count[qi] = kRoutePending;
                prm.ovf_list[atomicAdd(prm.ovf_count, 1)] = static_cast<int32_t>(qi);
            }
        }
        if (qi + q_stride >= prm.nq) {   // last query of this workgroup: nothing to tidy up, the stores drain on their own
            LZ_STAMP(6);
            LZ_STAMP(7);
            break;
        }
        __syncthreads();
        LZ_STAMP(6);
        // ---- clear exactly the slots this query used ---------------------------------------------------------------
        const int used = min(s_u, kLzEntries);
        for (int i = tid; i < used; i += nthreads) ht[ulist[i]] = kLzEmpty;
        if (kChk && prm.bin16) for (int i = tid; i < 1024; i += nthreads) bins[i] = 0;   // step 0 of the next query counts into it
        __syncthreads();
        LZ_STAMP(7);
    }
#undef LZ_STOP
#undef LZ_STAMP
#undef LZ_FOR_TUPLES
#undef LZ_INSERT
}

template <int kThreads, int kEnt, bool kChk, int kTD = 0, int kP = 0>
__global__ __launch_bounds__(kThreads, (kEnt <= 512 ? 6 : (kEnt <= 1024 ? 4 : 2))) void route_select_lazy_kernel(RouteParams prm) {
    extern __shared__ __align__(16) unsigned char smem[];
    route_lazy_run<kThreads, kEnt, kChk, kTD, kP>(prm, smem, static_cast<int64_t>(blockIdx.x), static_cast<int64_t>(gridDim.x), static_cast<int>(blockIdx.x));
}

}  // namespace fspann
