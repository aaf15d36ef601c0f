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
constexpr uint64_t kInvalidKey = ~0ull;

struct RefinePartial {  // one per (query, chunk, rank)
    uint64_t key;       // fp64 distance bits (non-negative => monotone as u64)
    int32_t pos;        // candidate position in F_q (stable tie-break)
    int32_t id;
};

template <typename T> struct VecOf;
template <> struct VecOf<float> { using type = float4; static constexpr int N = 4; };
template <> struct VecOf<double> { using type = double2; static constexpr int N = 2; };

template <typename T> __device__ __forceinline__ bool finite_t(T x) {
    return fabs(static_cast<double>(x)) <= 1.79769313486231570815e+308;
}

// TC = candidate dtype, TQ = query dtype, DC = dims per LDS tile, VEC = use 16-B loads.
template <typename TC, typename TQ, int DC, bool VEC>
__global__ __launch_bounds__(kRefRows) void refine_scan_kernel(
    const TQ* __restrict__ q, const TC* __restrict__ cand, int64_t B, int d, const int32_t* __restrict__ cand_ids,
    const int32_t* __restrict__ cand_count, int k, int nchunks, int32_t* __restrict__ out_ids,
    double* __restrict__ out_dist, int32_t* __restrict__ out_count, int32_t* __restrict__ scored,
    RefinePartial* __restrict__ partial, int32_t* __restrict__ partial_cnt) {
    constexpr int LD = kRefRows + 1;  // +1 word pad: conflict-light transposed writes
    extern __shared__ __align__(16) unsigned char smem[];
    double* qs = reinterpret_cast<double*>(smem);                     // [d]
    TC* tile = reinterpret_cast<TC*>(smem + static_cast<size_t>((d + 1) & ~1) * 8);  // [DC][LD]
    __shared__ uint64_t keys[kRefRows];
    __shared__ int s_qbad;
    __shared__ int s_nvalid;

    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x / nchunks;
    const int chunk = blockIdx.x - static_cast<int>(qi) * nchunks;
    const int cnt = min(static_cast<int64_t>(cand_count[qi]), B);
    const int r0 = chunk * kRefRows;
    const int nrows = max(0, min(kRefRows, cnt - r0));
    const TC* base = cand + (qi * B + r0) * static_cast<int64_t>(d);

    if (tid == 0) { s_qbad = 0; s_nvalid = 0; }
    __syncthreads();
    for (int i = tid; i < d; i += kRefRows) {
        const TQ v = q[qi * d + i];
        if (!finite_t(v)) s_qbad = 1;
        qs[i] = static_cast<double>(v);
    }

    using V = typename VecOf<TC>::type;
    constexpr int VN = VecOf<TC>::N;
    constexpr int VPR = DC / VN;                       // vectors per row per tile
    constexpr int NV = (kRefRows * VPR) / kRefRows;    // vectors per lane per tile (= VPR)
    V reg[NV];

    auto issue = [&](int c0) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const int v = tid + i * kRefRows;
                const int row = v / VPR, cv = v - row * VPR;
                const int col = c0 + cv * VN;
                if (row < nrows && col < d) reg[i] = *reinterpret_cast<const V*>(base + static_cast<int64_t>(row) * d + col);
            }
        }
    };
    auto commit = [&](int c0) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const int v = tid + i * kRefRows;
                const int row = v / VPR, cv = v - row * VPR;
                const int col = c0 + cv * VN;
                if (row < nrows && col < d) {
                    const TC* e = reinterpret_cast<const TC*>(&reg[i]);
#pragma unroll
                    for (int x = 0; x < VN; x++) tile[(cv * VN + x) * LD + row] = e[x];
                }
            }
        } else {
            for (int e = tid; e < kRefRows * DC; e += kRefRows) {
                const int row = e / DC, cc = e - row * DC;
                if (row < nrows && c0 + cc < d) tile[cc * LD + row] = base[static_cast<int64_t>(row) * d + c0 + cc];
            }
        }
    };

    double s = 0.0;
    bool ok = true;
    issue(0);
    for (int c0 = 0; c0 < d; c0 += DC) {
        __syncthreads();  // previous tile fully consumed (and qs visible on the first pass)
        commit(c0);
        if (c0 + DC < d) issue(c0 + DC);
        __syncthreads();
        if (tid < nrows) {
            const int dc = min(DC, d - c0);
#pragma unroll 8
            for (int kk = 0; kk < dc; kk++) {
                const TC x = tile[kk * LD + tid];
                ok = ok && finite_t(x);
                const double dd = qs[c0 + kk] - static_cast<double>(x);  // QSI.java:368
                const double sq = dd * dd;
                s = s + sq;                                              // QSI.java:369
            }
        }
    }
    const bool qbad = (s_qbad != 0);
    const bool valid = (tid < nrows) && ok && !qbad;
    uint64_t key = kInvalidKey;
    if (valid) key = static_cast<uint64_t>(__double_as_longlong(sqrt(s)));  // QSI.java:371
    keys[tid] = key;
    if (valid) atomicAdd(&s_nvalid, 1);
    __syncthreads();
    const int nvalid = s_nvalid;

    // stable rank among the chunk's valid rows
    int rank = 0;
    if (valid) {
        for (int j = 0; j < nrows; j++) {
            const uint64_t kj = keys[j];
            rank += (kj < key) || (kj == key && j < tid);
        }
    }
    const int eff = min(k, nvalid);
    if (nchunks == 1) {
        if (valid && rank < eff) {
            out_ids[qi * k + rank] = cand_ids[qi * B + r0 + tid];
            out_dist[qi * k + rank] = __longlong_as_double(static_cast<long long>(key));
        }
        for (int i = eff + tid; i < k; i += kRefRows) {
            out_ids[qi * k + i] = -1;
            out_dist[qi * k + i] = __longlong_as_double(0x7FF0000000000000LL);
        }
        if (tid == 0) {
            out_count[qi] = eff;
            if (scored) scored[qi] = nvalid;
        }
    } else {
        if (valid && rank < eff) {
            RefinePartial pp;
            pp.key = key;
            pp.pos = r0 + tid;
            pp.id = cand_ids[qi * B + r0 + tid];
            partial[(qi * nchunks + chunk) * k + rank] = pp;
        }
        if (tid == 0) {
            partial_cnt[(qi * nchunks + chunk) * 2 + 0] = eff;
            partial_cnt[(qi * nchunks + chunk) * 2 + 1] = nvalid;
        }
    }
}

// Merge of per-chunk sorted top-k lists (B > kRefRows).  Each list is sorted by
// (key, pos) and chunks hold disjoint, increasing pos ranges, so the global rank of
// an element is its own rank plus, per other chunk, an upper/lower bound.
__global__ __launch_bounds__(256) void refine_merge_kernel(const RefinePartial* __restrict__ partial,
                                                           const int32_t* __restrict__ partial_cnt, int nchunks, int k,
                                                           int32_t* __restrict__ out_ids, double* __restrict__ out_dist,
                                                           int32_t* __restrict__ out_count, int32_t* __restrict__ scored) {
    const int64_t qi = blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ int s_total, s_nvalid;
    if (tid == 0) {
        int t = 0, nv = 0;
        for (int c = 0; c < nchunks; c++) {
            t += partial_cnt[(qi * nchunks + c) * 2];
            nv += partial_cnt[(qi * nchunks + c) * 2 + 1];
        }
        s_total = t;
        s_nvalid = nv;
    }
    __syncthreads();
    const int eff = min(k, s_total);
    const int nelem = nchunks * k;
    for (int e = tid; e < nelem; e += blockDim.x) {
        const int c = e / k, rk = e - c * k;
        const int cc = partial_cnt[(qi * nchunks + c) * 2];
        if (rk >= cc) continue;
        const RefinePartial me = partial[(qi * nchunks + c) * k + rk];
        int rank = rk;
        for (int c2 = 0; c2 < nchunks && rank < eff; c2++) {
            if (c2 == c) continue;
            const int n2 = partial_cnt[(qi * nchunks + c2) * 2];
            const RefinePartial* lst = partial + (qi * nchunks + c2) * k;
            // c2 < c: count keys <= me.key (earlier positions win ties); c2 > c: keys < me.key
            int lo = 0, hi = n2;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const uint64_t km = lst[mid].key;
                const bool before = (c2 < c) ? (km <= me.key) : (km < me.key);
                if (before) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < eff) {
            out_ids[qi * k + rank] = me.id;
            out_dist[qi * k + rank] = __longlong_as_double(static_cast<long long>(me.key));
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
