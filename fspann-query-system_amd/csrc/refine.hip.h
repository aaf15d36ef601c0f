// refine.hip.h — Refine on gfx950: the B-bounded exact-L2 scan + top-K of
// QueryServiceImpl.search stage B/C (QSI:238-316), distance = QSI.l2 (QSI:364-372).
//
// The reference sums (q_i - v_i)^2 sequentially in fp64 and takes Math.sqrt; the
// order of the additions is part of the result, so a row is never split across
// lanes.  Instead the [rows x dims] candidate block is streamed from HBM with
// coalesced 16-byte loads (8 lanes cover one 128-byte row segment), transposed
// through LDS, and each lane then walks ITS row in dimension order in fp64 with
// contraction off: bit-identical to Java whenever the inputs are exactly
// representable (fvecs/bvecs data are), <= 1 ulp of fp64 otherwise.
//
// One workgroup = kRefRows candidate rows of one query.  Global loads for tile
// t+1 are issued into registers before tile t is consumed (register double
// buffering), so each CU keeps >= 2 tiles of HBM traffic in flight.
// Top-K: stable rank by (distance bits, candidate position) through an all-pairs
// LDS broadcast compare — no sort network, no barriers in the loop.
#pragma once
#include "fspann_common.h"

#pragma clang fp contract(off)

namespace fspann {

constexpr int kRefRows = 256;     // rows (= lanes) per workgroup
// Cache policy of the dense row stream (aux operand of the buffer loads): 2 = nt.  The rows are read once; streamed with the
// default policy they sweep Route's working set (id rows, inverse map, partition records: ~200 MB) out of the 256 MiB Infinity
// Cache every step.
#ifndef FSPANN_REFINE_LOAD_AUX
#define FSPANN_REFINE_LOAD_AUX 2
#endif
constexpr int kRefCountUnknown = -0x7FFFFFFF;
constexpr uint64_t kInvalidKey = ~0ull;

struct RefinePartial {  // one per (query, chunk, rank)
    uint64_t key;       // fp64 distance bits (non-negative => monotone as u64)
    int32_t pos;        // candidate position in F_q (stable tie-break)
    int32_t id;
};

template <typename T> struct VecOf;
// native clang vectors (plain SSA values: a HIP float4 staging array ends up in scratch)
typedef float fsp_f32x4 __attribute__((ext_vector_type(4)));
typedef double fsp_f64x2 __attribute__((ext_vector_type(2)));
template <> struct VecOf<float> { using type = fsp_f32x4; static constexpr int N = 4; };
template <> struct VecOf<double> { using type = fsp_f64x2; static constexpr int N = 2; };

template <typename T> __device__ __forceinline__ bool finite_t(T x) {
    return fabs(static_cast<double>(x)) <= 1.79769313486231570815e+308;
}

__device__ __forceinline__ double vcomp(fsp_f32x4 v, int e) { return static_cast<double>(v[e]); }
__device__ __forceinline__ double vcomp(fsp_f64x2 v, int e) { return v[e]; }

constexpr int kRefFilterMaxK = 32;   // top-k via per-wave k-th-smallest filter up to this k

// rank of (key, pos) among keys[0..n): #{j : key_j < key || (key_j == key && j < pos)}; keys are read two
// at a time (16-byte LDS loads); n is padded to an even count with kInvalidKey by the caller.
__device__ __forceinline__ int rank_among(const uint64_t* keys, int n, uint64_t key, int pos) {
    const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(keys);
    int rank = 0;
#pragma unroll 4
    for (int j = 0; j < n / 2; j++) {
        const ulonglong2 kk = k2[j];
        rank += (kk.x < key) || (kk.x == key && 2 * j < pos);
        rank += (kk.y < key) || (kk.y == key && 2 * j + 1 < pos);
    }
    return rank;
}

// TC = candidate dtype, TQ = query dtype, DC = dims per LDS tile, VEC = use 16-B loads.
// LDS tile is ROW-major with a 16-byte pad per row (pitch DC*sizeof(TC)+16): the 16-byte writes of
// 8 consecutive lanes fill one row segment, and lane r's 16-byte reads of ITS row hit banks
// (36 r + 4 kk) mod 64 -> conflict-free for ds_read_b128 / ds_write_b128 (MI355X_MICROARCH.md §LDS).
//
// GATHER = false: `cand` is the dense [nq][B][d] block of decrypted candidate rows (the host decrypt loop of
// QSI:238-271 produced it).  GATHER = true: `cand` is the device-resident plaintext store [store_n][d] and row j of
// query qi is store[cand_ids[qi*B + j]] — the rows are read from the store exactly once, no staging copy; an id
// outside [0, store_n) is a point that failed to load (QSI:252-256: skipped, not scored).
// Arguments of one refinement scan (one launch, or the refine role of tick_kernel).
template <typename TC, typename TQ>
struct RefineArgs {
    const TQ* q;
    const TC* cand;              // dense [nq][B][d] block, or the store [store_n][d] (GATHER)
    int64_t store_n, B;
    int d;
    const int32_t* cand_ids;
    const int32_t* cand_count;
    int k, nchunks;
    int32_t* out_ids;
    double* out_dist;
    int32_t* out_count;
    int32_t* scored;
    RefinePartial* partial;
    int32_t* partial_cnt;
    int npieces, cpp;            // running top-k (refine_topk_running): a query's chunks are cut into npieces runs of cpp consecutive chunks,
                                 //   one workgroup walks a run and keeps its best k in LDS; 0 / 0: one partial list per chunk
    long long* dbg;              // FSPANN_DEBUG_STAMPS builds: [grid][4 waves][16] wall_clock64 stamps of each workgroup's first unit (else unused)
};

// Stage C for one 256-row chunk (QSI:298-316): stable rank of the chunk's distances by (fp64 bits, candidate position),
// the first min(k, valid) written out (or handed to refine_merge_kernel as a partial list when B spans several chunks).
// `tile` = the workgroup's LDS tile, dead at this point: every wave keeps its scratch in its own 64 rows of it.
template <typename TC, typename TQ, int PITCH>
__device__ __forceinline__ void refine_topk_emit(const RefineArgs<TC, TQ>& a, TC* tile, const bool valid, const uint64_t key,
                                                 const int32_t my_id, const int64_t qi, const int chunk, const int r0,
                                                 long long* stamps = nullptr) {
    // debug builds: stamps go to LDS and are flushed by the caller after the unit (a global store here would sit in front of
    // the next s_waitcnt vmcnt(0) and the stamp would measure its own latency)
#ifdef FSPANN_DEBUG_STAMPS
    __shared__ long long s_em_stamps[kRefRows / 64][16];
#define EM_STAMP(i) do { if (stamps && (threadIdx.x & 63) == 0) s_em_stamps[threadIdx.x >> 6][(i)] = wall_clock64(); } while (0)
#else
#define EM_STAMP(i) do { (void)stamps; } while (0)
#endif
    const int k = a.k, nchunks = a.nchunks;
    int32_t* __restrict__ out_ids = a.out_ids;
    double* __restrict__ out_dist = a.out_dist;
    int32_t* __restrict__ out_count = a.out_count;
    int32_t* __restrict__ scored = a.scored;
    RefinePartial* __restrict__ partial = a.partial;
    int32_t* __restrict__ partial_cnt = a.partial_cnt;
    __shared__ uint64_t s_wcut[kRefRows / 64];
    __shared__ int s_wbase[kRefRows / 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- stable rank by (distance bits, candidate position) -------------------------------------
    // Scratch lives in each wave's OWN (now dead) 64 tile rows until the first barrier (other waves may still be reading
    // theirs): bytes [0, 512) = the wave's 64 keys; afterwards bytes [512, 1536) of region c hold entries [64 c, 64 c + 64)
    // of the chunk's survivor list (16 bytes each: key, id, candidate position).
    constexpr size_t kRegion = static_cast<size_t>(64) * PITCH * sizeof(TC);
    static_assert(kRegion >= 1536 && kRegion % 16 == 0, "a wave's tile rows hold its keys and a quarter of the survivor list");
    auto wave_scratch = [&](int w) { return reinterpret_cast<uint64_t*>(reinterpret_cast<unsigned char*>(tile) + static_cast<size_t>(w) * kRegion); };
    auto surv = [&](int i) { return reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(tile) + static_cast<size_t>(i >> 6) * kRegion + 512) + (i & 63); };
    uint64_t* wkeys = wave_scratch(wave);
    const bool filter = (k <= kRefFilterMaxK);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    wkeys[lane] = key;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int wvalid = __popcll(__ballot(valid));
    if (filter) {
        // All the cut has to be is an UPPER BOUND of the chunk's k-th smallest key; the survivors are ranked exactly below.
        // The wave bisects the top word of its keys (finite non-negative doubles: bit 31 clear; an invalid lane never counts):
        // T = the largest multiple of 2^kCutLow with fewer than k of the wave's top words below it, so at least k keys lie
        // below (T + 2^kCutLow) << 32.  One vector compare per step, the rest is scalar work — this epilogue is the tail of
        // the launch, every wave of a SIMD is in it at once and they share one vector pipe (an exact all-pairs rank of the
        // wave's 64 keys: ~400 vector instructions per wave, 1.5 us of the launch).
        // NOTE on the inline asm below and in the survivor masks: v_cmp_*_e64 writes its lane mask into an arbitrary scalar pair, and the
        // compiler's hazard recogniser does not see that write as a VALU write of an SGPR.  The masks may therefore only feed scalar ALU
        // and plain vector instructions (s_bcnt, mbcnt, shifts — what they feed today); used as a VMEM scalar operand or a readlane /
        // writelane selector they would need the 4-5 wait states of the gfx9 "VALU writes SGPR" hazards inserted by hand.
        // (compares through the icmp builtin: its result IS the lane mask in scalar registers — one vector instruction and
        // four scalar ones per candidate; __ballot of a bool costs two more vector instructions, and every hop between
        // the vector and scalar pipes is a pipeline latency on this dependent chain: 1.5 us of the launch's tail as 21
        // one-bit steps through __ballot, two bits per step here)
        constexpr int kCutLow = 10;
        const uint32_t hi = static_cast<uint32_t>(key >> 32);       // an invalid lane's key is all ones: never below a candidate
        auto below = [&](const uint32_t c) { return __popcll(__builtin_amdgcn_uicmp(hi, c, 36 /* ICMP_ULT */)); };
        uint32_t T = (below(1u << 30) < k) ? (1u << 30) : 0u;
#pragma unroll
        for (int b = 28; b >= kCutLow; b -= 2) {
            const uint32_t c1 = T | (1u << b), c2 = T | (2u << b), c3 = T | (3u << b);
            // the three compares of a step back to back, each into its own scalar pair: ONE trip from the vector to the scalar
            // pipe per step (left to the compiler they all go through VCC, one trip each: 60 of the chain's 90 pipe crossings)
            unsigned long long m1, m2, m3;
            asm volatile("v_cmp_gt_u32_e64 %0, %3, %6\n\tv_cmp_gt_u32_e64 %1, %4, %6\n\tv_cmp_gt_u32_e64 %2, %5, %6"
                         : "=&s"(m1), "=&s"(m2), "=&s"(m3) : "s"(c1), "s"(c2), "s"(c3), "v"(hi));
            const int n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
            T = (n3 < k) ? c3 : (n2 < k) ? c2 : (n1 < k) ? c1 : T;
        }
        static_assert(kCutLow % 2 == 0, "two bits per step from bit 29 down");
        const uint64_t wc = (wvalid >= k) ? ((static_cast<uint64_t>(T + (1u << kCutLow)) << 32) - 1ull) : kInvalidKey;
        if (lane == 0) s_wcut[wave] = wc;
    }
    if (lane == 0) s_wbase[wave] = wvalid;
    EM_STAMP(6);
    __syncthreads();   // (1) keys, cuts and valid counts of all waves
    EM_STAMP(7);
    int nvalid = 0;
    uint64_t cutk = kInvalidKey;
#pragma unroll
    for (int w = 0; w < kRefRows / 64; w++) {
        nvalid += s_wbase[w];
        if (filter) cutk = min(cutk, s_wcut[w]);
    }
    const int eff = min(k, nvalid);
    // a winner goes to its place in the result (or, when B spans several chunks, in this chunk's partial list)
    auto emit = [&](const int rank, const uint64_t wkey, const int32_t wid, const int32_t wpos) {
        if (nchunks == 1) {
            out_ids[qi * k + rank] = wid;
            out_dist[qi * k + rank] = __longlong_as_double(static_cast<long long>(wkey));
        } else {
            RefinePartial pp;
            pp.key = wkey;
            pp.pos = wpos;
            pp.id = wid;
            partial[(qi * nchunks + chunk) * k + rank] = pp;
        }
    };
    if (filter) {
        // Survivors = valid keys <= the tightest cut (>= eff of them, typically ~3 k).  Every wave derives the survivor masks
        // of ALL four waves from the keys in LDS (four loads and ballots — no second exchange of counts), so it knows where
        // its own survivors go in ONE dense list, in candidate-position order.
        unsigned long long bm_mine = 0;
        int mybase = 0, T = 0;
        const uint64_t cut_valid = min(cutk, kInvalidKey - 1);      // no cut at all (no wave holds k valid keys): every valid key
        static_assert(kRefRows / 64 == 4, "four waves");
        const uint64_t kw0 = wave_scratch(0)[lane], kw1 = wave_scratch(1)[lane], kw2 = wave_scratch(2)[lane], kw3 = wave_scratch(3)[lane];
        unsigned long long bmw[4];
        asm volatile("v_cmp_le_u64_e64 %0, %4, %8\n\tv_cmp_le_u64_e64 %1, %5, %8\n\tv_cmp_le_u64_e64 %2, %6, %8\n\tv_cmp_le_u64_e64 %3, %7, %8"
                     : "=&s"(bmw[0]), "=&s"(bmw[1]), "=&s"(bmw[2]), "=&s"(bmw[3]) : "v"(kw0), "v"(kw1), "v"(kw2), "v"(kw3), "s"(cut_valid));
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w == wave) { mybase = T; bm_mine = bmw[w]; }
            T += __popcll(bmw[w]);
        }
        if ((bm_mine >> lane) & 1ull) {
            // (mbcnt = set bits of the mask below this lane: no lane mask to build, hoist and keep alive through the stream)
            const int at = mybase + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(bm_mine >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(bm_mine), 0u)));
            *surv(at) = make_uint4(static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), static_cast<uint32_t>(my_id), static_cast<uint32_t>(r0 + tid));
        }
        EM_STAMP(8);
        __syncthreads();   // (2) the survivor list
        EM_STAMP(9);
        // Exact rank of survivor i = #{j : (key_j, j) < (key_i, i)}, the T x T pairs spread over the WHOLE workgroup: 2^lgP
        // consecutive lanes share one i and split the j (T ~ 30: eight lanes per survivor, four pairs each), their partial
        // counts meet in a few xor-shuffles — a few dozen vector instructions per wave where every wave ranking its own
        // survivors against all lists took several hundred and a chain of dependent LDS reads.
        const int lgT = (T <= 1) ? 0 : 32 - __clz(T - 1);     // T <= 2^lgT <= 256
        const int lgP = min(6, 8 - lgT);
        const int i = tid >> lgP, part = tid & ((1 << lgP) - 1);
        int r = 0;
        if (i < T) {
            const uint2 mk = *reinterpret_cast<const uint2*>(surv(i));
            const uint64_t ki = static_cast<uint64_t>(mk.x) | (static_cast<uint64_t>(mk.y) << 32);
            for (int j0 = part; j0 < T; j0 += (4 << lgP)) {         // four list entries per trip, their loads issued together
                uint2 o[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + (u << lgP);
                    o[u] = *reinterpret_cast<const uint2*>(surv(j < T ? j : i));
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + (u << lgP);
                    const uint64_t kj = static_cast<uint64_t>(o[u].x) | (static_cast<uint64_t>(o[u].y) << 32);
                    r += (j < T) && ((kj < ki) || (kj == ki && j < i));
                }
            }
        }
        // the 2^lgP partial counts of a survivor meet: inside a row of 16 lanes through DPP (no LDS round trip), beyond through shuffles
        if (lgP >= 1) r += __builtin_amdgcn_update_dpp(0, r, 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]
        if (lgP >= 2) r += __builtin_amdgcn_update_dpp(0, r, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
        if (lgP >= 3) r += __builtin_amdgcn_update_dpp(0, r, 0x141, 0xF, 0xF, false);    // row_half_mirror
        if (lgP >= 4) r += __builtin_amdgcn_update_dpp(0, r, 0x140, 0xF, 0xF, false);    // row_mirror
        if (lgP >= 5) r += __shfl_xor(r, 16);
        if (lgP >= 6) r += __shfl_xor(r, 32);
        EM_STAMP(10);
        if (i < T && part == 0 && r < eff) {
            const uint4 me = *surv(i);
            emit(r, static_cast<uint64_t>(me.x) | (static_cast<uint64_t>(me.y) << 32), static_cast<int32_t>(me.z), static_cast<int32_t>(me.w));
        }
    } else if (valid) {
        int rank = 0;
#pragma unroll
        for (int w = 0; w < kRefRows / 64; w++) {
            const ulonglong2* l2 = reinterpret_cast<const ulonglong2*>(wave_scratch(w));
            if (w < wave) { for (int j = 0; j < 32; j++) { const ulonglong2 kk = l2[j]; rank += (kk.x <= key && kk.x != kInvalidKey) + (kk.y <= key && kk.y != kInvalidKey); } }
            else if (w > wave) { for (int j = 0; j < 32; j++) { const ulonglong2 kk = l2[j]; rank += (kk.x < key) + (kk.y < key); } }
            else rank += rank_among(wkeys, 64, key, lane);
        }
        if (rank < eff) emit(rank, key, my_id, r0 + tid);
    }

    if (nchunks == 1) {
        for (int i = eff + tid; i < k; i += kRefRows) {
            out_ids[qi * k + i] = -1;
            out_dist[qi * k + i] = __longlong_as_double(0x7FF0000000000000LL);
        }
        if (tid == 0) {
            out_count[qi] = eff;
            if (scored) scored[qi] = nvalid;
        }
    } else if (tid == 0) {
        partial_cnt[(qi * nchunks + chunk) * 2 + 0] = eff;
        partial_cnt[(qi * nchunks + chunk) * 2 + 1] = nvalid;
    }
    EM_STAMP(11);
#ifdef FSPANN_DEBUG_STAMPS
    if (stamps && (threadIdx.x & 63) == 0) for (int i = 6; i <= 11; i++) stamps[i] = s_em_stamps[threadIdx.x >> 6][i];
#endif
#undef EM_STAMP
}

// Stage C for a RUN of consecutive chunks of one query walked by ONE workgroup (long candidate lists: B in the thousands, k = 100):
// the best k of the chunks seen so far stay in LDS, sorted by (distance bits, candidate position); a chunk contributes only the
// keys BELOW the current k-th (later positions lose ties) — a handful per chunk once the list has filled — and they are merged
// by rank: an old entry moves up by the number of newcomers in front of it, a newcomer's place is its bound in the old list
// plus the newcomers in front of it.  With one list per chunk (refine_topk_emit) SIFT_P10_HIGH wrote 86 lists of 100 per query,
// ranked all 256 keys of every chunk against each other and merged the lists in a second kernel (0.33 ms per 1024 queries).
// first / last: the run's first and last chunk; at `last` the list is written out — the final result when the run is the whole
// query (npieces == 1: no merge kernel at all), else the partial list of piece `piece`.
constexpr int kRunMaxK = 128;
template <typename TC, typename TQ, int PITCH>
__device__ __forceinline__ void refine_topk_running(const RefineArgs<TC, TQ>& a, TC* tile, const bool valid, const uint64_t key, const int32_t my_id,
                                                    const int64_t qi, const int piece, const int pos, const bool first, const bool last) {
    __shared__ uint4 s_best[kRunMaxK];             // (key lo, key hi, id, candidate position)
    __shared__ int s_nbest, s_nvalid_run, s_wcnt[kRefRows / 64], s_wval[kRefRows / 64];
    const int k = a.k;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr size_t kRegion = static_cast<size_t>(64) * PITCH * sizeof(TC);
    static_assert(kRegion >= 64 * 16 && kRegion % 16 == 0, "a wave's tile rows hold its survivors");
    auto wlist = [&](int w) { return reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(tile) + static_cast<size_t>(w) * kRegion); };
    auto key_of = [](const uint4 e) { return static_cast<uint64_t>(e.x) | (static_cast<uint64_t>(e.y) << 32); };
    const int L = first ? 0 : s_nbest;             // (the previous chunk's update is behind the unit-end barrier)
    uint64_t thr = kInvalidKey;                    // list not full: every valid key enters (valid keys are finite: below all-ones)
    if (L == k) thr = key_of(s_best[k - 1]);
    const bool sv = valid && key < thr;            // strictly: an equal key at a later position ranks behind the k-th
    const unsigned long long bm = __ballot(sv), bv = __ballot(valid);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();               // this wave's tile rows are dead (other waves may still be reading theirs)
    if (sv) {
        const int at = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(bm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(bm), 0u)));
        wlist(wave)[at] = make_uint4(static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), static_cast<uint32_t>(my_id), static_cast<uint32_t>(pos));
    }
    if (lane == 0) { s_wcnt[wave] = __popcll(bm); s_wval[wave] = __popcll(bv); }
    __syncthreads();   // (1) survivors and counts of all waves
    static_assert(kRefRows / 64 == 4, "four waves");
    const int c0 = s_wcnt[0], c1 = s_wcnt[1], c2 = s_wcnt[2], c3 = s_wcnt[3];
    const int S = c0 + c1 + c2 + c3;
    const int nv = s_wval[0] + s_wval[1] + s_wval[2] + s_wval[3];
    if (S > 0) {                                   // block-uniform
        // every entry of old list + newcomers finds its place (at most two entries per thread: L <= 128, S <= 256)
        uint4 me[2];
        int rk[2] = {0x7FFFFFFF, 0x7FFFFFFF};
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {
            const int t = tid + t2 * kRefRows;
            if (t >= L + S) continue;
            if (t < L) {
                me[t2] = s_best[t];
                const uint64_t mk = key_of(me[t2]);
                int r = t;
                for (int w = 0; w < 4; w++) {
                    const int cw = s_wcnt[w];
                    const uint4* wl = wlist(w);
                    for (int j = 0; j < cw; j++) r += key_of(wl[j]) < mk;           // an old entry wins ties (earlier position)
                }
                rk[t2] = r;
            } else {
                int j = t - L, w = 0;
                if (j >= c0) { j -= c0; w = 1; if (j >= c1) { j -= c1; w = 2; if (j >= c2) { j -= c2; w = 3; } } }
                me[t2] = wlist(w)[j];
                const uint64_t mk = key_of(me[t2]);
                int lo = 0, hi = L;                                                 // old entries with key <= mine
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (key_of(s_best[mid]) <= mk) lo = mid + 1; else hi = mid; }
                int r = lo;
                for (int w2 = 0; w2 < 4; w2++) {
                    const int cw = s_wcnt[w2];
                    const uint4* wl = wlist(w2);
                    for (int j2 = 0; j2 < cw; j2++) {
                        const uint4 o = wl[j2];
                        const uint64_t ok_ = key_of(o);
                        r += (ok_ < mk) || (ok_ == mk && o.w < me[t2].w);
                    }
                }
                rk[t2] = r;
            }
        }
        __syncthreads();   // (2) every read of the old list and of the newcomers is done
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++)
            if (rk[t2] < k) s_best[rk[t2]] = me[t2];
    }
    if (tid == 0) {
        s_nbest = min(k, L + S);
        s_nvalid_run = (first ? 0 : s_nvalid_run) + nv;
    }
    if (last) {
        __syncthreads();   // (3) the list is complete
        const int nb = s_nbest;
        if (a.npieces == 1) {
            for (int i = tid; i < k; i += kRefRows) {
                if (i < nb) {
                    const uint4 e = s_best[i];
                    a.out_ids[qi * k + i] = static_cast<int32_t>(e.z);
                    a.out_dist[qi * k + i] = __longlong_as_double(static_cast<long long>(key_of(e)));
                } else {
                    a.out_ids[qi * k + i] = -1;
                    a.out_dist[qi * k + i] = __longlong_as_double(0x7FF0000000000000LL);
                }
            }
            if (tid == 0) {
                a.out_count[qi] = nb;
                if (a.scored) a.scored[qi] = s_nvalid_run;
            }
        } else {
            const int64_t li = qi * a.npieces + piece;
            for (int i = tid; i < nb; i += kRefRows) {
                const uint4 e = s_best[i];
                RefinePartial pp;
                pp.key = key_of(e);
                pp.pos = static_cast<int32_t>(e.w);
                pp.id = static_cast<int32_t>(e.z);
                a.partial[li * k + i] = pp;
            }
            if (tid == 0) {
                a.partial_cnt[li * 2 + 0] = nb;
                a.partial_cnt[li * 2 + 1] = s_nvalid_run;
            }
        }
    }
}

// One workgroup (kRefRows threads) = one 256-row chunk of one query: block `bidx` of nq * nchunks.  `smem` = dynamic LDS.
template <typename TC, typename TQ, int DC, bool VEC, bool GATHER>
__device__ __forceinline__ void refine_scan_block(const RefineArgs<TC, TQ>& a, unsigned char* smem, const int64_t bidx,
                                                  const int cnt_known = kRefCountUnknown) {
    const TQ* __restrict__ q = a.q;
    const TC* __restrict__ cand = a.cand;
    const int64_t store_n = a.store_n, B = a.B;
    const int d = a.d, nchunks = a.nchunks;
    const int32_t* __restrict__ cand_ids = a.cand_ids;
    const int32_t* __restrict__ cand_count = a.cand_count;
    using V = typename VecOf<TC>::type;
    constexpr int VN = VecOf<TC>::N;
    constexpr int PITCH = VEC ? DC + VN : DC + 1;   // elements per LDS row
    constexpr int VPR = DC / VN;                    // 16-byte vectors per row per tile
    // LDS budget: <= 40 KB per workgroup so that FOUR workgroups (all 1024 of a 1024-query batch) are resident per
    // CU; the top-k scratch (keys, surv) therefore aliases the tile, which is dead once the scan loop is done, and
    // the query vector is not staged at all: its address is wave-uniform, so it is read through the scalar cache.
    TC* tile = reinterpret_cast<TC*>(smem);                                              // [kRefRows][PITCH]
    const TQ* __restrict__ qrow = q + (bidx / nchunks) * d;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t qi = bidx / nchunks;
    const int chunk = static_cast<int>(bidx - qi * nchunks);
    const int r0 = chunk * kRefRows;
    const TC* base = GATHER ? cand : cand + (qi * B + r0) * static_cast<int64_t>(d);
    // The candidate ids do not depend on cand_count: they are requested first (all slots of this chunk that exist in
    // the [nq][B] id array), so the count, the ids and the query check are ONE global round trip before the rows.
    const int rows_here = static_cast<int>(min(static_cast<int64_t>(kRefRows), B - r0));
    const int32_t my_id_raw = (tid < rows_here) ? cand_ids[qi * B + r0 + tid] : -1;
    // register double buffering: tile t+1 is in flight (global -> VGPR) while tile t is consumed from LDS
    V reg[VPR];
    // source row of each of this lane's 16-byte slots: the block-local row (dense) or the store row (gather), -1 = none
    int32_t srow[VPR];
    if constexpr (VEC && GATHER) {
#pragma unroll
        for (int i = 0; i < VPR; i++) {
            const int row = wave * 64 + (lane + i * 64) / VPR;
            srow[i] = (row < rows_here) ? cand_ids[qi * B + r0 + row] : -1;
        }
    }
    // QSI.java:137-140: a non-finite query gives an empty result.  Every wave looks at the whole query itself.
    bool qnf = false;
    for (int i = lane; i < d; i += 64) qnf = qnf || !__builtin_isfinite(qrow[i]);
    const bool qbad = __any(qnf);
    // cnt_known: the caller has read cand_count[qi] itself (tick_kernel re-reads it after redoing a PENDING query's Route)
    const int cnt = static_cast<int>(min(static_cast<int64_t>(cnt_known != kRefCountUnknown ? cnt_known : cand_count[qi]), B));
    const int nrows = max(0, min(kRefRows, cnt - r0));
    // candidate id of this lane's row (the epilogue has no dependent global load)
    const int32_t my_id = (tid < nrows) ? my_id_raw : -1;
    if constexpr (VEC) {
#pragma unroll
        for (int i = 0; i < VPR; i++) {
            const int row = wave * 64 + (lane + i * 64) / VPR;
            if constexpr (GATHER) srow[i] = (row < nrows && srow[i] >= 0 && srow[i] < store_n) ? srow[i] : -1;
            else srow[i] = (row < nrows) ? row : -1;
        }
    }
#define FSP_ISSUE(C0)                                                                                              \
    if constexpr (VEC) {                                                                                           \
        _Pragma("unroll") for (int i = 0; i < VPR; i++) {                                                          \
            const int col = (C0) + ((lane + i * 64) % VPR) * VN;                                                   \
            if (srow[i] >= 0 && col < d) reg[i] = *reinterpret_cast<const V*>(base + static_cast<int64_t>(srow[i]) * d + col); \
        }                                                                                                          \
    }
    double s = 0.0;
    bool ok = GATHER ? (my_id >= 0 && my_id < store_n) : true;
    FSP_ISSUE(0)
    // Each wave stages and consumes ITS OWN 64 rows: no workgroup barrier in the loop, the four waves drift apart
    // and overlap each other's load / LDS / fp64 phases.  LDS operations of one wave complete in program order.
#define FSP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
    for (int c0 = 0; c0 < d; c0 += DC) {
        FSP_WAVE_SYNC();  // previous tile fully consumed by this wave
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < VPR; i++) {
                const int v = lane + i * 64;
                const int row = wave * 64 + v / VPR, cv = v % VPR;
                const int col = c0 + cv * VN;
                if (srow[i] >= 0 && col < d) *reinterpret_cast<V*>(tile + row * PITCH + cv * VN) = reg[i];
            }
        } else {
            for (int e = lane; e < 64 * DC; e += 64) {
                const int row = wave * 64 + e / DC, cc = e % DC;
                if (row < nrows && c0 + cc < d) {
                    int64_t sr = row;
                    if constexpr (GATHER) {
                        const int32_t id = cand_ids[qi * B + r0 + row];
                        sr = (id >= 0 && id < store_n) ? id : -1;
                    }
                    if (sr >= 0) tile[row * PITCH + cc] = base[sr * d + c0 + cc];
                }
            }
        }
        if (c0 + DC < d) { FSP_ISSUE(c0 + DC) }
        FSP_WAVE_SYNC();
        if (tid < nrows) {
            const int dc = min(DC, d - c0);
            const TC* myrow = tile + tid * PITCH;
            if constexpr (VEC) {
#pragma unroll 4
                for (int kk = 0; kk < dc; kk += VN) {
                    const V xv = *reinterpret_cast<const V*>(myrow + kk);
#pragma unroll
                    for (int e = 0; e < VN; e += 2) {
                        const double q0 = static_cast<double>(qrow[c0 + kk + e]);      // uniform address -> scalar load
                        const double q1 = static_cast<double>(qrow[c0 + kk + e + 1]);
                        ok = ok && __builtin_isfinite(xv[e]) && __builtin_isfinite(xv[e + 1]);   // v_cmp_class on the raw element
                        const double x0 = vcomp(xv, e), x1 = vcomp(xv, e + 1);   // exact widening
                        const double d0 = q0 - x0;                               // QSI.java:368
                        const double p0 = d0 * d0;
                        s = s + p0;                                              // QSI.java:369 (in order)
                        const double d1 = q1 - x1;
                        const double p1 = d1 * d1;
                        s = s + p1;
                    }
                }
            } else {
                for (int kk = 0; kk < dc; kk++) {
                    const TC x = myrow[kk];
                    ok = ok && finite_t(x);
                    const double dd = static_cast<double>(qrow[c0 + kk]) - static_cast<double>(x);
                    const double sq = dd * dd;
                    s = s + sq;
                }
            }
        }
    }
#undef FSP_ISSUE
#undef FSP_WAVE_SYNC
    const bool valid = (tid < nrows) && ok && !qbad;
    uint64_t key = kInvalidKey;
    if (valid) key = static_cast<uint64_t>(__double_as_longlong(sqrt(s)));  // QSI.java:371
    refine_topk_emit<TC, TQ, PITCH>(a, tile, valid, key, my_id, qi, chunk, r0);
}

template <typename TC, typename TQ, int DC, bool VEC, bool GATHER>
__global__ __launch_bounds__(kRefRows, (DC * sizeof(TC) <= 128 ? 4 : (DC * sizeof(TC) <= 256 ? 2 : 1))) void refine_scan_kernel(RefineArgs<TC, TQ> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    refine_scan_block<TC, TQ, DC, VEC, GATHER>(a, smem, static_cast<int64_t>(blockIdx.x));
}

// ------------------------------------------------------------------------------------------------------------------
// The same scan as a STREAM: one workgroup walks the units (query, 256-row chunk) u = wg, wg + nwg, wg + 2 nwg, ... and
// keeps the row loads of the next tile in flight at all times — across unit boundaries, so the loads of the next
// query are under way while this query's top-K is ranked and written.
//
// Why: with one workgroup per query (refine_scan_block) every workgroup of the launch is in the same phase at the same
// time — all wait for their ids, all stream, all rank — and HBM idles during the first and the last phase; and inside
// tick_kernel each of those workgroups holds one of the CU's four slots for ~25 us, most of it waiting.  As a stream, a
// quarter of the workgroups (one or two per CU) keep HBM just as busy (one 32 KB tile in flight per workgroup), the
// bubbles overlap with streaming, and the other slots are free for the latency-bound Route workgroups.
//
// Numerics are those of refine_scan_block: each lane walks ITS row in dimension order in fp64 (QSI:364-372), the top-K
// is refine_topk_emit.  VEC layout only (d % (16 / sizeof(TC)) == 0, 16-byte aligned rows): the host falls back otherwise.
// counts_fresh: read cand_count with a device-scope atomic load (tick_kernel: the count of a PENDING query was rewritten
// by this very workgroup a moment ago; a plain load could hit a stale scalar / L1 line).
// `fix` (tick.hip.h: refine_stream_fix_kernel): called for a query whose Route the bounded select handed over (count = PENDING) and
// finished here by the full select before its rows are scored; RefineNoFix = nothing to finish.
struct RefineNoFix {
    static constexpr bool enabled = false;
    __device__ __forceinline__ void operator()(int64_t) const {}
};
// kMulti (dense blocks, several chunks per query, 32 < k <= kRunMaxK): the workgroup's units are RUNS of consecutive chunks of one
// query (item = query * npieces + piece, items wg, wg + nwg, ...; a.cpp chunks per run) and the top-K is refine_topk_running.
template <typename TC, typename TQ, int DC, bool GATHER, class FixFn = RefineNoFix, bool kMulti = false>
__device__ __forceinline__ void refine_stream_run(const RefineArgs<TC, TQ>& a, unsigned char* smem, const int64_t wg, const int64_t nwg,
                                                  const int64_t nq, const bool counts_fresh, const FixFn fix = FixFn()) {
    static_assert(!(kMulti && (GATHER || FixFn::enabled)), "runs of chunks: dense blocks, no hand-over");
    using V = typename VecOf<TC>::type;
    constexpr int VN = VecOf<TC>::N;
    constexpr int PITCH = DC + VN;
    constexpr int VPR = DC / VN;
    // fp32 rows against an fp32 query: |q - x| < 2^129, so the fp64 sum of squares cannot overflow, and a NaN or an infinity
    // in the row always reaches the sum — "every element finite" (QSI.isValid, QSI:407-413) is "the sum is finite", one test
    // per row instead of one per element (a sixth of the scan's vector instructions).
    constexpr bool kSumTellsFinite = (sizeof(TC) == 4 && sizeof(TQ) == 4);
    const TC* __restrict__ cand = a.cand;
    const int64_t store_n = a.store_n, B = a.B;
    const int d = a.d, nchunks = a.nchunks;
    // Everything the first tile's addresses need, requested from the kernel-argument segment in ONE round: left alone, the
    // compiler fetches the block pointer and d in a second round of scalar loads behind the unit arithmetic — a second
    // memory round trip in front of the first row load, with HBM idle.
    asm volatile("" :: "s"(cand), "s"(B), "s"(d), "s"(nchunks), "s"(nq));
    asm volatile("" :: "s"(a.q), "s"(a.cand_ids), "s"(a.cand_count), "s"(a.k), "s"(a.out_ids), "s"(a.out_dist), "s"(a.out_count), "s"(a.scored));
    asm volatile("" :: "s"(a.partial), "s"(a.partial_cnt), "s"(a.dbg), "s"(a.store_n));
    const int32_t* __restrict__ cand_ids = a.cand_ids;
    TC* tile = reinterpret_cast<TC*>(smem);                                              // [kRefRows][PITCH]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // a scalar: what derives from it is scalar work
    const int ntile = (d + DC - 1) / DC;
    const int64_t nunits = kMulti ? nq * a.npieces : nq * nchunks;      // kMulti: ITEMS (runs of chunks)
    // position in the workgroup's sequence of chunks (kMulti): run `item`, its query, first chunk and length, the chunk inside it
    struct RunPos { int64_t item, qi; int cc, nch, chunk0; };
    auto run_set = [&](RunPos& rp, const int64_t item) {
        rp.item = item; rp.cc = 0;
        const uint32_t it_ = static_cast<uint32_t>(min(item, nunits - 1));           // past the end: the last run's numbers (never consumed)
        const uint32_t qq = it_ / static_cast<uint32_t>(a.npieces);
        rp.qi = qq;
        rp.chunk0 = static_cast<int>(it_ - qq * static_cast<uint32_t>(a.npieces)) * a.cpp;
        rp.nch = min(a.cpp, nchunks - rp.chunk0);
    };
    RunPos ip{}, cp{};                                     // issue side, consume side
    if constexpr (kMulti) { run_set(ip, wg); run_set(cp, wg); }
    const int slot_row = wave * 64 + lane / VPR;          // + i * (64 / VPR): row of this lane's i-th 16-byte slot
    const int slot_col = (lane % VPR) * VN;               // column of the slot inside a tile

    // ---- issue side: position (unit, tile) one tile ahead of the consume side -----------------------------------
    int64_t iu = wg;                                       // unit being requested
    int it = 0;                                            // its next tile
    // Every load of the stream is UNCONDITIONAL (a slot without a row — beyond the block's rows, an id outside the store,
    // a column beyond d in a partial last tile — re-reads a row / column that exists and its data is never used): with
    // predicated loads the compiler waits for ALL outstanding loads at every tile instead of only the older register set,
    // and the prefetch is gone (measured: 4.7 us per tile instead of ~1.3).
    int32_t isrc[GATHER ? VPR : 1];                        // gather: store row of each slot for unit iu (clamped into the store)
    int irows = 1;                                         // dense: rows of unit iu that exist in the block
    // unit -> (query, first row): no division at all for one chunk per query, a 32-bit one otherwise (the launcher keeps
    // nunits below 2^31) — a 64-bit division is ~100 scalar instructions, and two of them sat in front of the first load
    auto split_unit = [&](const int64_t uu, int64_t& qi, int& r0) {
        if (nchunks == 1) { qi = uu; r0 = 0; return; }
        const uint32_t qq = static_cast<uint32_t>(uu) / static_cast<uint32_t>(nchunks);
        qi = qq;
        r0 = static_cast<int>(static_cast<uint32_t>(uu) - qq * static_cast<uint32_t>(nchunks)) * kRefRows;
    };
    auto load_sources = [&](const int64_t u) {
        const int64_t uu = min(u, nunits - 1);             // past the end: addresses of the last unit (never consumed)
        int64_t qi; int r0;
        split_unit(uu, qi, r0);
        const int rows_here = static_cast<int>(min(static_cast<int64_t>(kRefRows), B - r0));
        irows = rows_here;
        if constexpr (GATHER) {
            if (u >= nunits) {                                      // past the last unit: every lane re-reads row 0 (one cache line per load)
#pragma unroll
                for (int i = 0; i < VPR; i++) isrc[i] = 0;
                return;
            }
#pragma unroll
            for (int i = 0; i < VPR; i++) {
                const int row = min(slot_row + i * (64 / VPR), rows_here - 1);
                const int32_t id = cand_ids[qi * B + r0 + row];
                isrc[i] = (id >= 0 && id < store_n) ? id : 0;      // a point that failed to load (QSI:252-256): row 0 is read, never scored
            }
        }
    };
    auto unit_base = [&](const int64_t u) -> const TC* {
        if constexpr (GATHER) return cand;
        if constexpr (kMulti) {
            const int r0 = (ip.chunk0 + ip.cc) * kRefRows;
            irows = static_cast<int>(min(static_cast<int64_t>(kRefRows), B - r0));
            return cand + (ip.qi * B + r0) * static_cast<int64_t>(d);
        }
        const int64_t uu = min(u, nunits - 1);
        int64_t qi; int r0;
        split_unit(uu, qi, r0);
        return cand + (qi * B + r0) * static_cast<int64_t>(d);
    };
    const TC* ibase = unit_base(iu);
    if constexpr (!kMulti) load_sources(iu);
    // Dense blocks are read through a BUFFER resource per unit (base = the unit's first row, extent = its rows): the address
    // of slot i is one 32-bit lane offset plus a wave-uniform scalar offset (no 64-bit address arithmetic, no address
    // registers per slot), and a slot beyond the unit's rows is answered with zeros by the range check instead of a clamp.
    // Past the last unit the stream keeps issuing (the loads stay unconditional, see below) but the resource's extent is zero:
    // every such load is out of range and is answered with zeros without touching memory.
    auto unit_rsrc = [&]() {
        const int64_t extent = (iu < nunits) ? static_cast<int64_t>(irows) * d * static_cast<int64_t>(sizeof(TC)) : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<TC*>(ibase), 0, static_cast<int>(extent), 0x00020000);
    };
    __amdgpu_buffer_rsrc_t irsrc = unit_rsrc();
    const int slot_off = (slot_row * d + slot_col) * static_cast<int>(sizeof(TC));      // byte offset of slot 0 in tile 0
    const int slot_step = (64 / VPR) * d * static_cast<int>(sizeof(TC));                // bytes from slot i to slot i + 1
#define FSP_STREAM_ISSUE(REG)                                                                                       \
    do {                                                                                                            \
        if constexpr (GATHER) {                                                                                     \
            int col_ = it * DC + slot_col;                                                                          \
            col_ = (col_ < d) ? col_ : 0;                                                                           \
            _Pragma("unroll") for (int i = 0; i < VPR; i++)                                                         \
                REG[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(ibase + static_cast<int64_t>(isrc[GATHER ? i : 0]) * d + col_)); \
        } else {                                                                                                    \
            const int voff_ = slot_off + it * DC * static_cast<int>(sizeof(TC));                                    \
            _Pragma("unroll") for (int i = 0; i < VPR; i++)                                                         \
                REG[i] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(irsrc, voff_, i * slot_step, FSPANN_REFINE_LOAD_AUX)); \
        }                                                                                                           \
        if (++it == ntile) {                                                                                        \
            it = 0;                                                                                                 \
            if constexpr (kMulti) {                                                                                 \
                if (++ip.cc == ip.nch) { run_set(ip, ip.item + nwg); iu = ip.item; }                                \
                ibase = unit_base(iu);                                                                              \
            } else {                                                                                                \
                iu += nwg;                                                                                          \
                ibase = unit_base(iu);                                                                              \
                load_sources(iu);                                                                                   \
            }                                                                                                       \
            if constexpr (!GATHER) irsrc = unit_rsrc();                                                             \
        }                                                                                                           \
    } while (0)
#define FSP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

#ifdef FSPANN_DEBUG_STAMPS
    __shared__ long long s_rs_stamps[kRefRows / 64][16];
#define RS_STAMP(i) do { if (a.dbg && lane == 0 && first_unit) s_rs_stamps[wave][(i)] = wall_clock64(); } while (0)
#else
#define RS_STAMP(i) do { } while (0)
#endif
    bool first_unit = true; (void)first_unit;
    RS_STAMP(0);
    // ONE tile in flight per wave (requested before the previous one is consumed).  Two were in flight until round 3: the same
    // bandwidth alone (a pure-load kernel reads 6.2-6.3 TB/s either way, tools/ubench/scan_pattern.hip), but 64 MB of queued
    // requests in front of the memory channels instead of 32 — every access of a kernel running BESIDE the scan (Route's chain of
    // dependent reads) waited behind them.  Same-box A/B: step 46.6 -> 45.2 us, the scan alone 27.4 -> 26.4 us (and 30 registers less).
    V regA[VPR];
    FSP_STREAM_ISSUE(regA);
    RS_STAMP(1);
    if constexpr (FixFn::enabled) {
        // The first tile is under way; now look at the counts of this workgroup's queries (one 256-row chunk per query
        // here: unit == query).  PENDING = handed over by the bounded select: finish its Route with the full select first —
        // rare, so the stream simply starts again afterwards (the tiles requested above are dropped: nothing of them is live
        // across the full select, which needs the registers).
        // (the counts through the CONSTANT address space: scalar loads, which do not queue behind the tile loads just issued — as
        // vector loads they waited for the tile; the Route launch that wrote them finished before this kernel started)
        typedef const int32_t __attribute__((address_space(4)))* const_cnt_t;
        const const_cnt_t cc = (const_cnt_t)a.cand_count;
        bool any = false;
        for (int64_t u = wg; u < nunits; u += nwg) any = any || (cc[u] == -2 /* kRoutePending */);
        if (any) {
            for (int64_t u = wg; u < nunits; u += nwg)
                if (cc[u] == -2) fix(u);
            iu = wg; it = 0;
            ibase = unit_base(iu);
            load_sources(iu);
            if constexpr (!GATHER) irsrc = unit_rsrc();
            FSP_STREAM_ISSUE(regA);
        }
    }

    // ---- consume side ------------------------------------------------------------------------------------------------
    for (int64_t u = wg; u < nunits;) {
        int64_t qi; int r0;
        if constexpr (kMulti) { qi = cp.qi; r0 = (cp.chunk0 + cp.cc) * kRefRows; }
        else split_unit(u, qi, r0);
        const int chunk = r0 / kRefRows;
        // the query row through the CONSTANT address space: uniform loads from it are scalar loads whatever else the
        // enclosing kernel does (see encode_exact_block); the query batch is an input, nothing writes it
        typedef const TQ __attribute__((address_space(4)))* const_row_t;
        const const_row_t qrow = (const_row_t)(a.q + qi * d);
        const int rows_here = static_cast<int>(min(static_cast<int64_t>(kRefRows), B - r0));
        const int32_t my_id_raw = (tid < rows_here) ? cand_ids[qi * B + r0 + tid] : -1;
        int cnt_raw;
        if (counts_fresh) cnt_raw = __hip_atomic_load(const_cast<int32_t*>(a.cand_count) + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else cnt_raw = a.cand_count[qi];
        // QSI.java:137-140: a non-finite query gives an empty result.  Every wave looks at the whole query itself — unless the
        // sum tells (fp32 query, fp32 rows): a NaN or an infinity in the query reaches EVERY row's sum, every row is invalid,
        // and "no valid row" is exactly the empty result with scored = 0 (two vector loads and a ballot less in front of the
        // stream: 0.6 us of the launch, same-box A/B).
        bool qnf = false;
        if constexpr (!kSumTellsFinite) for (int i = lane; i < d; i += 64) qnf = qnf || !__builtin_isfinite(qrow[i]);
        // The count, this lane's id and the query check are REQUESTED here but first USED after the last tile: the scan
        // below runs over every row the block holds (tid < rows_here) and the count only decides, in the epilogue, which
        // rows are scored — so no tile waits for this round trip.
        double s = 0.0;
        bool ok = true;

#define FSP_STREAM_TILE(REG, C0)                                                                                    \
        do {                                                                                                        \
            FSP_WAVE_SYNC();  /* previous tile fully consumed by this wave */                                       \
            _Pragma("unroll") for (int i = 0; i < VPR; i++)   /* unconditional: rows / columns without data are never read */ \
                *reinterpret_cast<V*>(tile + (slot_row + i * (64 / VPR)) * PITCH + slot_col) = REG[i];               \
            FSP_STREAM_ISSUE(REG);                                                                                  \
            FSP_WAVE_SYNC();                                                                                        \
            if (tid < rows_here) {                                                                                  \
                const int dc = min(DC, d - (C0));                                                                   \
                const TC* myrow = tile + tid * PITCH;                                                               \
                _Pragma("unroll 4") for (int kk = 0; kk < dc; kk += VN) {                                           \
                    const V xv = *reinterpret_cast<const V*>(myrow + kk);                                           \
                    _Pragma("unroll") for (int e = 0; e < VN; e += 2) {                                             \
                        const double q0 = static_cast<double>(qrow[(C0) + kk + e]);      /* uniform address -> scalar load */ \
                        const double q1 = static_cast<double>(qrow[(C0) + kk + e + 1]);                             \
                        if constexpr (!kSumTellsFinite) ok = ok && __builtin_isfinite(xv[e]) && __builtin_isfinite(xv[e + 1]); \
                        const double x0 = vcomp(xv, e), x1 = vcomp(xv, e + 1);   /* exact widening */                \
                        const double d0 = q0 - x0;                               /* QSI.java:368 */                  \
                        const double p0 = d0 * d0;                                                                  \
                        s = s + p0;                                              /* QSI.java:369 (in order) */       \
                        const double d1 = q1 - x1;                                                                  \
                        const double p1 = d1 * d1;                                                                  \
                        s = s + p1;                                                                                 \
                    }                                                                                               \
                }                                                                                                   \
            }                                                                                                       \
        } while (0)

        // tiles of this unit; the load of the NEXT tile of the stream (this unit's or the next unit's) is issued inside
        // FSP_STREAM_TILE, right after this one has been written to LDS
        for (int t = 0; t < ntile; t++) {
            FSP_STREAM_TILE(regA, t * DC);
            if (t == 0) RS_STAMP(2);
            if (t == 1) RS_STAMP(3);
        }
        RS_STAMP(4);
        if constexpr (kSumTellsFinite) ok = __builtin_isfinite(s);
        const bool qbad = kSumTellsFinite ? false : __any(qnf);
        const int cnt = static_cast<int>(min(static_cast<int64_t>(cnt_raw), B));
        const int nrows = max(0, min(kRefRows, cnt - r0));
        const int32_t my_id = (tid < nrows) ? my_id_raw : -1;
        if (GATHER) ok = ok && (my_id >= 0 && my_id < store_n);
        const bool valid = (tid < nrows) && ok && !qbad;
        uint64_t key = kInvalidKey;
        if (valid) key = static_cast<uint64_t>(__double_as_longlong(sqrt(s)));  // QSI.java:371
        RS_STAMP(5);
        if constexpr (kMulti) {
            refine_topk_running<TC, TQ, PITCH>(a, tile, valid, key, my_id, qi, static_cast<int>(cp.item - qi * a.npieces), r0 + tid, cp.cc == 0, cp.cc == cp.nch - 1);
        } else {
#ifdef FSPANN_DEBUG_STAMPS
        refine_topk_emit<TC, TQ, PITCH>(a, tile, valid, key, my_id, qi, chunk, r0, (a.dbg && first_unit) ? a.dbg + (wg * 4 + wave) * 16 : nullptr);
#else
        refine_topk_emit<TC, TQ, PITCH>(a, tile, valid, key, my_id, qi, chunk, r0);
#endif
        }
        RS_STAMP(12);
#ifdef FSPANN_DEBUG_STAMPS
        if (a.dbg && lane == 0 && first_unit) { for (int i = 0; i <= 5; i++) a.dbg[(wg * 4 + wave) * 16 + i] = s_rs_stamps[wave][i]; a.dbg[(wg * 4 + wave) * 16 + 12] = s_rs_stamps[wave][12]; }
#endif
        __syncthreads();       // every wave has finished reading the other waves' scratch before the tile is written again
        first_unit = false;
        if constexpr (kMulti) { if (++cp.cc == cp.nch) { run_set(cp, cp.item + nwg); u = cp.item; } }
        else u += nwg;
    }
#undef RS_STAMP
#undef FSP_STREAM_TILE
#undef FSP_STREAM_ISSUE
#undef FSP_WAVE_SYNC
}

template <typename TC, typename TQ, int DC, bool GATHER, bool kMulti = false>
__global__ __launch_bounds__(kRefRows, (GATHER ? 2 : 4)) void refine_stream_kernel(RefineArgs<TC, TQ> a, int64_t nq) {
    extern __shared__ __align__(16) unsigned char smem[];
    refine_stream_run<TC, TQ, DC, GATHER, RefineNoFix, kMulti>(a, smem, static_cast<int64_t>(blockIdx.x), static_cast<int64_t>(gridDim.x), nq, false);
}

// Merge of per-chunk sorted top-k lists (B > kRefRows).  Each list is sorted by
// (key, pos) and chunks hold disjoint, increasing pos ranges, so the global rank of
// an element is its own rank plus, per other chunk, an upper/lower bound.
// KEYS_IN_LDS: the keys of all lists (nchunks * k * 8 bytes) and the list lengths are staged in LDS first, so the binary
// searches run at LDS latency instead of L2 latency (the shipped profiles: 32 / 86 lists of 100).  The host picks the
// variant by size.
// A cut first: with j = ceil(k / nchunks), the largest j-th key over the lists has at least k keys at or below it (j from
// every list), so only each list's PREFIX up to that key can hold a winner — ~3 k elements instead of nchunks * k, and
// the searches run over prefixes of a few entries (SIFT_P10_HIGH: 86 lists x 100 -> ~4 per list).
template <bool KEYS_IN_LDS>
__global__ __launch_bounds__(256) void refine_merge_kernel(const RefinePartial* __restrict__ partial,
                                                           const int32_t* __restrict__ partial_cnt, int nchunks, int k,
                                                           int32_t* __restrict__ out_ids, double* __restrict__ out_dist,
                                                           int32_t* __restrict__ out_count, int32_t* __restrict__ scored) {
    extern __shared__ __align__(16) unsigned char merge_smem[];
    uint64_t* s_keys = reinterpret_cast<uint64_t*>(merge_smem);                       // [nchunks][k]            (KEYS_IN_LDS)
    int32_t* s_pref = reinterpret_cast<int32_t*>(merge_smem + (KEYS_IN_LDS ? static_cast<size_t>(nchunks) * k * 8 : 0));   // [nchunks] prefix lengths
    const int64_t qi = blockIdx.x;
    const int tid = threadIdx.x;
    const int nelem = nchunks * k;
    __shared__ int s_total, s_nvalid, s_short;
    __shared__ unsigned long long s_cut;
    auto key_at = [&](int c, int i) -> uint64_t { return KEYS_IN_LDS ? s_keys[c * k + i] : partial[(qi * nchunks + c) * k + i].key; };
    if (tid == 0) { s_total = 0; s_nvalid = 0; s_short = 0; s_cut = 0ull; }
    if constexpr (KEYS_IN_LDS)
        for (int e = tid; e < nelem; e += blockDim.x) s_keys[e] = partial[qi * nelem + e].key;     // entries beyond a list's length are never read
    __syncthreads();
    // Only lists that HOLD j keys can vouch for j keys at or below the cut.  A query with fewer candidates than B leaves trailing
    // lists empty or short (the common partial-count case): with m lists of at least ceil(k / nchunks) keys, j = ceil(k / m) is tried —
    // the lists holding j vouch, and the cut stands when they vouch for k between them; the shorter lists are searched like the others.
    __shared__ int s_m;
    const int j0 = (k + nchunks - 1) / nchunks;
    if (tid == 0) s_m = 0;
    __syncthreads();
    for (int c = tid; c < nchunks; c += blockDim.x) {
        const int cc = partial_cnt[(qi * nchunks + c) * 2];
        s_pref[c] = cc;                                            // the list's length until the cut is known
        atomicAdd(&s_total, cc);
        atomicAdd(&s_nvalid, partial_cnt[(qi * nchunks + c) * 2 + 1]);
        if (cc >= j0) atomicAdd(&s_m, 1);
    }
    __syncthreads();
    const int j = s_m > 0 ? (k + s_m - 1) / s_m : 0;
    for (int c = tid; c < nchunks; c += blockDim.x) {
        const int cc = s_pref[c];
        if (j > 0 && cc >= j) { atomicAdd(&s_short, 1); atomicMax(&s_cut, static_cast<unsigned long long>(key_at(c, j - 1))); }   // s_short: lists that vouch
    }
    __syncthreads();
    const int eff = min(k, s_total);
    if (j > 0 && static_cast<long long>(s_short) * j >= k) {        // the vouching lists hold >= k keys <= cut between them
        const uint64_t cut = s_cut;
        for (int c = tid; c < nchunks; c += blockDim.x) {
            int lo = 0, hi = s_pref[c];                            // first index with key > cut
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (key_at(c, mid) <= cut) lo = mid + 1; else hi = mid; }
            s_pref[c] = lo;
        }
    }
    __syncthreads();
    for (int e = tid; e < nelem; e += blockDim.x) {
        const int c = e / k, rk = e - c * k;
        if (rk >= s_pref[c]) continue;
        const uint64_t mykey = key_at(c, rk);
        int rank = rk;
        for (int c2 = 0; c2 < nchunks && rank < eff; c2++) {
            if (c2 == c) continue;
            // c2 < c: count keys <= mykey (earlier positions win ties); c2 > c: keys < mykey
            int lo = 0, hi = s_pref[c2];
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const uint64_t km = key_at(c2, mid);
                const bool before = (c2 < c) ? (km <= mykey) : (km < mykey);
                if (before) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < eff) {
            out_ids[qi * k + rank] = partial[qi * nelem + e].id;
            out_dist[qi * k + rank] = __longlong_as_double(static_cast<long long>(mykey));
        }
    }
    for (int i = eff + tid; i < k; i += blockDim.x) {
        out_ids[qi * k + i] = -1;
        out_dist[qi * k + i] = __longlong_as_double(0x7FF0000000000000LL);
    }
    if (tid == 0) {
        out_count[qi] = eff;
        if (scored) scored[qi] = s_nvalid;
    }
}

// Plaintext-store gather (TEST/BENCH stand-in for host load+decrypt): one wave-
// instruction moves 1 KiB; rows are copied with 16-byte lanes when aligned.
template <typename T>
__global__ __launch_bounds__(256) void store_gather_kernel(const T* __restrict__ store, int d,
                                                           const int32_t* __restrict__ sel_ids,
                                                           const int32_t* __restrict__ sel_count, int64_t B, int64_t nq,
                                                           T* __restrict__ out, int vec_ok) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * (blockDim.x / 32) + (threadIdx.x / 32);
    const int lane = threadIdx.x & 31;
    if (row >= nq * B) return;
    const int64_t qi = row / B;
    const int j = static_cast<int>(row - qi * B);
    if (j >= sel_count[qi]) return;
    const int32_t id = sel_ids[row];
    if (id < 0) return;
    const T* src = store + static_cast<int64_t>(id) * d;
    T* dst = out + row * d;
    if (vec_ok) {
        using V = typename VecOf<T>::type;
        constexpr int VN = VecOf<T>::N;
        const int nv = d / VN;
        for (int i = lane; i < nv; i += 32) reinterpret_cast<V*>(dst)[i] = reinterpret_cast<const V*>(src)[i];
    } else {
        for (int i = lane; i < d; i += 32) dst[i] = src[i];
    }
}

}  // namespace fspann
