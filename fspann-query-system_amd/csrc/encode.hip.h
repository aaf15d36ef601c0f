// encode.hip.h — TokenGen math on gfx950: Coding.H + Coding.C for every (t,d).
//
// Reference (restated, not copied): idx/Coding.java:250-258 (H), :285-301 (C),
// :349-353 (dot), :355-361 (requireVector).  The reference accumulates the dot
// product sequentially in fp64 with no FMA; this kernel does exactly that on the
// fp64 VALU (one lane per projection, contraction off), so the integer hashes and
// therefore the codes are bit-identical.
//
// Layout: alphaT[dim][P] (P = T*D*m projections, projection index fastest) so a
// wave reads 64 consecutive doubles per dimension (512 B, coalesced); the QB query
// vectors of a block sit in LDS and are broadcast to all lanes.
#pragma once
#include <type_traits>
#include "fspann_common.h"

#pragma clang fp contract(off)

namespace fspann {

constexpr int kEncThreads = 256;
constexpr int kEncDC = 256;  // dims staged per LDS pass

__device__ __forceinline__ int32_t java_d2i(double x) {
    // Java (int) cast of a double: NaN -> 0, saturating.
    if (x != x) return 0;
    if (x >= 2147483647.0) return 2147483647;
    if (x <= -2147483648.0) return (-2147483647 - 1);
    return static_cast<int32_t>(x);
}

// Arguments of one exact-coding launch (or of the encode role of tick_kernel).
template <typename TIn>
struct EncodeArgs {
    const TIn* q;
    int64_t nq;
    int d;
    const double* alphaT;
    const double* r;
    const double* omega;
    int P, m, lambda, W, TD, tdPerBlock;
    uint64_t* codes;
    int32_t* hashes;
    int32_t* bad;
    double* proj;
    const unsigned long long* only_if_over;   // fallback launch behind the MFMA path: runs only when its re-check list overflowed
    unsigned long long over_cap;
    long long* dbg;                            // FSPANN_DEBUG_STAMPS builds: [grid][16] wall_clock64 stamps per workgroup (else unused)
};

// One workgroup (kEncThreads threads): QB query vectors x tdPerBlock tables; block (bx, by) of ceil(nq / QB) x ceil(TD / tdPerBlock).
// lds = (QB * kEncThreads + QB) int32 of LDS scratch (static in encode_exact_kernel, part of the dynamic LDS in tick_kernel).
template <typename TIn, int QB>
__device__ __forceinline__ void encode_exact_block(const EncodeArgs<TIn>& a, const int bx, const int by, int32_t* lds) {
    const TIn* __restrict__ q = a.q;
    const int64_t nq = a.nq;
    const int d = a.d, P = a.P, m = a.m, lambda = a.lambda, W = a.W, TD = a.TD, tdPerBlock = a.tdPerBlock;
    const double* __restrict__ alphaT = a.alphaT;
    const double* __restrict__ r = a.r;
    const double* __restrict__ omega = a.omega;
    uint64_t* __restrict__ codes = a.codes;
    int32_t* __restrict__ hashes = a.hashes;
    int32_t* __restrict__ bad = a.bad;
    double* __restrict__ proj = a.proj;
    if (a.only_if_over && *a.only_if_over <= a.over_cap) return;
#ifdef FSPANN_DEBUG_STAMPS
#define ENC_STAMP(i, dep) do { if (a.dbg && threadIdx.x == 0 && by == 0 && (dep) != -0x7654321) a.dbg[bx * 16 + (i)] = wall_clock64(); } while (0)
#else
#define ENC_STAMP(i, dep) do { } while (0)
#endif
    ENC_STAMP(0, 0);
    int32_t* Hs = lds;                       // [QB * kEncThreads]
    int* badq = lds + QB * kEncThreads;      // [QB]

    const int tid = threadIdx.x;
    const int64_t q0 = static_cast<int64_t>(bx) * QB;
    const int td0 = by * tdPerBlock;
    const int tdn = min(tdPerBlock, TD - td0);
    const int nproj = tdn * m;
    const bool active = tid < nproj;
    const int p = td0 * m + tid;

    // NaN / Inf in a query vector (Coding.java:356-361): every thread fetches its share of the QB rows NOW and looks at it
    // after the projection loop, so this round trip is hidden behind the loop (the sums simply carry a NaN)
    constexpr int kChk = 4;                       // elements per thread kept in registers; longer rows are checked in place
    TIn chk[kChk];
#pragma unroll
    for (int u = 0; u < kChk; u++) {
        const int idx = tid + u * kEncThreads;
        const int qq = idx / d;
        chk[u] = (idx < QB * d && q0 + qq < nq) ? q[(q0 + qq) * d + (idx - qq * d)] : TIn(0);
    }

    double acc[QB];
#pragma unroll
    for (int i = 0; i < QB; i++) acc[i] = 0.0;

    // The query elements are the same for every lane (one lane = one projection): their addresses are wave-uniform,
    // so they come through the scalar cache (s_load) and feed the fp64 multiply as scalar operands — no LDS staging,
    // no barrier in the loop.  alpha is streamed coalesced from alphaT (64 consecutive doubles per wave and dimension).
    // The rows are read through the CONSTANT address space: a uniform load from it is always a scalar load.  (Through the
    // plain global pointer the compiler must first prove that nothing in the kernel can write the row; it can in
    // encode_exact_kernel, it cannot once this function is one role of tick_kernel — the loads then become vector loads, each
    // alpha load waits for them, and a workgroup takes 41 us instead of 12.)  The query batch is an input: nothing writes it.
    typedef const TIn __attribute__((address_space(4)))* const_row_t;
    const_row_t qrow[QB];
#pragma unroll
    for (int qq = 0; qq < QB; qq++) qrow[qq] = (const_row_t)(q + min(q0 + qq, nq - 1) * d);   // rows past nq: computed, never stored
    ENC_STAMP(1, 0);
    if (active) {
        // alpha_j is streamed eight dimensions at a time, the NEXT eight requested before the current eight are used, and a
        // scheduling barrier keeps the compiler from sinking the loads to their uses (inside tick_kernel, whose route role
        // sits at the register ceiling, it otherwise emits load / wait / load / wait: eight dependent round trips per block).
        const double* ap = alphaT + p;
        constexpr int U = 8;
        const int dU = d & ~(U - 1);               // whole blocks of eight dimensions (no guards inside: the query elements of a
        if (dU > 0) {                              // block stay ONE 32-byte scalar load per query)
            double cur[U], nxt[U];
#pragma unroll
            for (int u = 0; u < U; u++) cur[u] = ap[static_cast<int64_t>(u) * P];
            for (int i = 0; i < dU; i += U) {
                const int inx = (i + U < dU) ? i + U : i;          // last block: re-request the current one (values unused)
#pragma unroll
                for (int u = 0; u < U; u++) nxt[u] = ap[static_cast<int64_t>(inx + u) * P];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < U; u++) {
#pragma unroll
                    for (int qq = 0; qq < QB; qq++) {
                        const double prod = static_cast<double>(qrow[qq][i + u]) * cur[u];  // acc += a[i]*b[i], Coding.java:351
                        acc[qq] = acc[qq] + prod;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++) cur[u] = nxt[u];
            }
        }
        for (int i = dU; i < d; i++) {
            const double a1 = ap[static_cast<int64_t>(i) * P];
#pragma unroll
            for (int qq = 0; qq < QB; qq++) {
                const double prod = static_cast<double>(qrow[qq][i]) * a1;
                acc[qq] = acc[qq] + prod;
            }
        }
    }

    // (no LDS store ahead of the projection loop: with one, the loop's query loads stop being scalar loads and every
    // alpha load waits for the previous one — measured 41 us per workgroup instead of 12 inside tick_kernel)
    ENC_STAMP(2, static_cast<int>(acc[0]));        // projection loop done
    if (tid < QB) badq[tid] = 0;
    __syncthreads();                                  // badq[] is zero
    ENC_STAMP(3, 0);
#pragma unroll
    for (int u = 0; u < kChk; u++) {
        const int idx = tid + u * kEncThreads;
        if (idx < QB * d && !(fabs(static_cast<double>(chk[u])) <= 1.79769313486231570815e+308)) atomicOr(&badq[idx / d], 1);
    }
    for (int idx = tid + kChk * kEncThreads; idx < QB * d; idx += kEncThreads) {
        const int qq = idx / d, i = idx - qq * d;
        const int64_t qi = q0 + qq;
        if (qi < nq && !(fabs(static_cast<double>(q[qi * d + i])) <= 1.79769313486231570815e+308)) atomicOr(&badq[qq], 1);
    }
    if (active) {
        const double rr = r[p], ww = omega[p];
#pragma unroll
        for (int qq = 0; qq < QB; qq++) {
            const double y = acc[qq] + rr;                        // Coding.java:254
            const int32_t h = java_d2i(floor(y / ww));            // Coding.java:255
            Hs[qq * kEncThreads + tid] = h;
            const int64_t qi = q0 + qq;
            if (hashes && qi < nq) hashes[qi * P + p] = h;
            if (proj && qi < nq) proj[qi * P + p] = acc[qq];  // raw dot(v, alpha_j), Coding.java:212
        }
    }
    __syncthreads();
    ENC_STAMP(4, 0);                                  // hashes in LDS

    // Coding.C: bit pos = (lambda-1-i)*m + j  <-  bit i of (h_j ^ 0x80000000)
    const int bitsTotal = m * lambda;
    if ((64 % m) == 0) {
        // m divides the wave: the m lanes of one table are adjacent, so plane i of a table's code is an m-bit field of
        // one ballot.  Lane j < W of the group assembles word j from the lambda fields (a field never straddles words).
        const int lane = tid & 63;
        const int g0 = (lane / m) * m, j = lane - g0;
        const int tdl = tid / m;
        const uint64_t fmask = (m == 64) ? ~0ull : ((1ull << m) - 1ull);
#pragma unroll
        for (int qq = 0; qq < QB; qq++) {
            const uint32_t hj = active ? (static_cast<uint32_t>(Hs[qq * kEncThreads + tid]) ^ 0x80000000u) : 0u;
            uint64_t word = 0;
            for (int i = 0; i < lambda; i++) {
                const unsigned long long bm = __ballot((hj >> (i & 31)) & 1u);
                const int pos0 = (lambda - 1 - i) * m;
                if ((pos0 >> 6) == j) word |= ((bm >> g0) & fmask) << (pos0 & 63);
            }
            const int64_t qi = q0 + qq;
            if (active && j < W && qi < nq) codes[(qi * TD + td0 + tdl) * W + j] = word;
        }
    } else {
        const int nwords = QB * tdn * W;
        for (int wi = tid; wi < nwords; wi += kEncThreads) {
            const int qq = wi / (tdn * W);
            const int rem = wi - qq * (tdn * W);
            const int tdl = rem / W, w = rem - tdl * W;
            const int64_t qi = q0 + qq;
            if (qi >= nq) continue;
            uint64_t word = 0;
            const int pos0 = w * 64;
            const int pos1 = min(bitsTotal, pos0 + 64);
            for (int pos = pos0; pos < pos1; pos++) {
                const int plane = pos / m;            // 0 .. lambda-1, MSB plane first
                const int j = pos - plane * m;
                const int i = lambda - 1 - plane;
                const uint32_t hj = static_cast<uint32_t>(Hs[qq * kEncThreads + tdl * m + j]) ^ 0x80000000u;
                word |= static_cast<uint64_t>((hj >> (i & 31)) & 1u) << (pos - pos0);
            }
            codes[(qi * TD + td0 + tdl) * W + w] = word;
        }
    }
    if (bad && by == 0 && tid < QB && q0 + tid < nq) bad[q0 + tid] = badq[tid];
    ENC_STAMP(5, 0);
#undef ENC_STAMP
}

template <typename TIn, int QB>
__global__ __launch_bounds__(kEncThreads) void encode_exact_kernel(EncodeArgs<TIn> a) {
    __shared__ int32_t lds[QB * kEncThreads + QB];
    encode_exact_block<TIn, QB>(a, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), lds);
}

}  // namespace fspann

// =====================================================================================
// MFMA fast path (SURVEY §7 hard part 1, plan (b)): y32 = V · A^T as an exact-f32 MFMA GEMM
// (v_mfma_f32_32x32x2_f32), then h = floor((y + r)/omega) is accepted only when y is provably
// on the same side of every bucket boundary as the reference's fp64 sum; the remaining
// (query, projection) pairs are recomputed by encode_fix_kernel with the exact fp64 chain.
// Results are therefore bit-identical to encode_exact_kernel by construction.
//
// Error bound: |y32 - y_ref| <= (d + 4) * 2^-24 * sum_i |v_i alpha_i| <= (d + 4) * 2^-24 * ||v||_2
// (alpha rows are L2-normalised, idx/Coding.java:149-150; the factor covers the f32 rounding of alpha
// and of an fp64 input, the k-ordered f32 fma chain and the f32 product roundings), inflated by 1.01.
// =====================================================================================
#pragma clang fp contract(off)
namespace fspann {

typedef float fsp_acc16 __attribute__((ext_vector_type(16)));
typedef float fsp_f4 __attribute__((ext_vector_type(4)));
constexpr int kMfmaKT = 32;        // K tile
// Block tile = (32 RT) queries x (128 CT) projections, four waves; wave w owns columns [32 CT w, 32 CT (w + 1)) = RT x CT MFMA tiles
// of 32 x 32.  RT = CT = 2 (64 x 256) for bulk coding — the index build, batches of thousands: the V tile is read once per 256
// projections; RT = CT = 1 (32 x 128) for a query batch of a few hundred to a few thousand rows, where the large tile would leave
// most CUs without a block (BASELINE config #3's shard: 512 queries x 256 projections = 8 blocks) and one wave would run 4 x 480
// dependent MFMAs instead of 480.
// Workgroup barrier that waits for THIS wave's LDS traffic only.  __syncthreads() also waits for every outstanding global load
// (s_waitcnt vmcnt(0) in front of s_barrier), which is exactly what a k loop with the next tile's loads in flight must not do.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int mfma_tile_q(int RT) { return 32 * RT; }
constexpr int mfma_tile_p(int CT) { return 128 * CT; }

// One block = 64 vectors x 256 projections (the V tile is read once per 256 projections: with the old 64 x 64 tile every vector
// row was fetched P / 64 times).  Epilogue: quantise (floor((y + r) / omega), Coding.java:254-255) AND bit-pack (Coding.C,
// :285-301) in place — the bits of every pair whose fp32 result is provably on the right side of its bucket edges are OR-ed
// straight into the (pre-zeroed) code words, one ballot per bit plane; the other pairs go to the fix list and encode_fix_kernel
// ORs THEIR bits after the exact fp64 chain.  No int32 H round trip through HBM, no pack kernel (hashes != null: H is written too).
template <typename TIn, int RT, int CT>
__global__ __launch_bounds__(256, 2) void encode_mfma_kernel(
    const TIn* __restrict__ q, int64_t nq, int d, const float* __restrict__ alphaT32 /*[d][P]*/,
    const double* __restrict__ r, const double* __restrict__ omega, int P, int m, int lambda, int W, int TD, int32_t* __restrict__ hashes,
    unsigned long long* __restrict__ codes, int32_t* __restrict__ bad, int64_t* __restrict__ fix_list, int64_t fix_cap,
    unsigned long long* __restrict__ fix_count, double alpha_norm_max) {
    constexpr int kMfmaTileQ = mfma_tile_q(RT), kMfmaTileP = mfma_tile_p(CT);
    __shared__ float Vs[kMfmaTileQ][kMfmaKT + 1];
    __shared__ __align__(16) float As[kMfmaKT][kMfmaTileP];
    __shared__ double rnorm2[kMfmaTileQ];
    __shared__ int badrow[kMfmaTileQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q0 = static_cast<int64_t>(blockIdx.x) * kMfmaTileQ;
    const int p0 = blockIdx.y * kMfmaTileP;
    if (tid < kMfmaTileQ) { rnorm2[tid] = 0.0; badrow[tid] = 0; }
    fsp_acc16 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; a++)
#pragma unroll
        for (int b = 0; b < CT; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.0f;
    constexpr int kVRows = kMfmaTileQ / 8;          // V rows per thread (thread t: column t & 31 of rows (t >> 5) + 8 i)
    constexpr int kAVecs = kMfmaKT * kMfmaTileP / 4 / 256;   // 16-byte pieces of the A tile per thread
    double nrm[kVRows];
#pragma unroll
    for (int i = 0; i < kVRows; i++) nrm[i] = 0.0;
    __syncthreads();
    const bool a_vec = ((P & 3) == 0);

    auto mfma_tile = [&]() {
#pragma unroll
        for (int kk = 0; kk < kMfmaKT; kk += 2) {
            float av[RT], bv[CT];
#pragma unroll
            for (int a = 0; a < RT; a++) av[a] = Vs[32 * a + (lane & 31)][kk + (lane >> 5)];
#pragma unroll
            for (int b = 0; b < CT; b++) bv[b] = As[kk + (lane >> 5)][wave * (32 * CT) + 32 * b + (lane & 31)];
#pragma unroll
            for (int a = 0; a < RT; a++)
#pragma unroll
                for (int b = 0; b < CT; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    };
    constexpr bool kAhead = (RT == 1 && CT == 1);
    if constexpr (kAhead) {
        // Small tile: kSets register sets in rotation keep the NEXT kSets - 1 tiles in flight: a set is written to LDS, refilled with the
        // tile kSets ahead, and its tile multiplied.  (Loaded and used inside the same trip, a block waited for a round trip to memory per
        // tile — with one block per CU nothing else hides it: 24-30 tiles x ~2.5 us = 65-100 us for a 512-1 024 row batch at d = 768 /
        // 960; a tile's 16 MFMAs are ~0.5 us, so one tile ahead is not enough.)
        // Every load is UNCONDITIONAL (an element outside the batch, d or P is read from a clamped address and replaced by zero when the
        // tile is written to LDS): behind a branch the compiler waits for each load before it issues the next (the ISA had a
        // `s_waitcnt vmcnt(0)` behind every guarded load); and the loop's barriers wait for LDS only (lds_barrier) — __syncthreads would
        // wait for the tiles in flight at the end of every trip.
        constexpr int kSets = 3;
        double vreg[kSets][kVRows];
        fsp_f4 areg[kSets][kAVecs];
        auto fetch = [&](auto set_c, const int k0) {
            constexpr int st = decltype(set_c)::value;
#pragma unroll
            for (int i = 0; i < kVRows; i++) {
                const int row = (tid >> 5) + 8 * i, col = tid & 31;
                vreg[st][i] = static_cast<double>(q[min(q0 + row, nq - 1) * d + min(k0 + col, d - 1)]);
            }
#pragma unroll
            for (int i = 0; i < kAVecs; i++) {
                const int e = (tid + i * 256) * 4;
                const int kk = e / kMfmaTileP, pc = e % kMfmaTileP;
                if (a_vec) {                           // uniform: P % 4 == 0, so a 16-byte piece lies wholly inside or wholly outside P
                    areg[st][i] = *reinterpret_cast<const fsp_f4*>(alphaT32 + static_cast<int64_t>(min(k0 + kk, d - 1)) * P + min(p0 + pc, P - 4));
                } else {
                    fsp_f4 a = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (k0 + kk < d) {
                        const float* src = alphaT32 + static_cast<int64_t>(k0 + kk) * P + p0 + pc;
#pragma unroll
                        for (int u = 0; u < 4; u++) if (p0 + pc + u < P) a[u] = src[u];
                    }
                    areg[st][i] = a;
                }
            }
        };
        auto tile_from = [&](auto set_c, const int k0) {
            constexpr int st = decltype(set_c)::value;
#pragma unroll
            for (int i = 0; i < kVRows; i++) {
                const int row = (tid >> 5) + 8 * i, col = tid & 31;
                const double v = (q0 + row < nq && k0 + col < d) ? vreg[st][i] : 0.0;
                if (!(fabs(v) <= 1.79769313486231570815e+308)) badrow[row] = 1;
                Vs[row][col] = static_cast<float>(v);
                nrm[i] += v * v;
            }
#pragma unroll
            for (int i = 0; i < kAVecs; i++) {
                const int e = (tid + i * 256) * 4;
                const int kk = e / kMfmaTileP, pc = e % kMfmaTileP;
                const fsp_f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                *reinterpret_cast<fsp_f4*>(&As[kk][pc]) = (!a_vec || (k0 + kk < d && p0 + pc < P)) ? areg[st][i] : zero;
            }
            lds_barrier();
            if (k0 + kSets * kMfmaKT < d) fetch(set_c, k0 + kSets * kMfmaKT);      // uniform: the set just emptied takes the tile kSets ahead
            mfma_tile();
            lds_barrier();
        };
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
        fetch(S0{}, 0);
        if (kMfmaKT < d) fetch(S1{}, kMfmaKT);
        if (2 * kMfmaKT < d) fetch(S2{}, 2 * kMfmaKT);
        for (int k0 = 0; k0 < d; k0 += kSets * kMfmaKT) {                           // uniform trips
            tile_from(S0{}, k0);
            if (k0 + kMfmaKT < d) tile_from(S1{}, k0 + kMfmaKT);
            if (k0 + 2 * kMfmaKT < d) tile_from(S2{}, k0 + 2 * kMfmaKT);
        }
    } else {
        // Bulk tile: straight from memory into LDS, trip by trip — its blocks hide each other's loads, and registers for a tile in flight
        // cost it the second block per CU (502 -> 609 us per 262 144 rows).
        for (int k0 = 0; k0 < d; k0 += kMfmaKT) {
            // V tile: kMfmaTileQ rows x 32 k (thread t: column t&31 of rows (t>>5) + 8 i); A tile: 32 k x kMfmaTileP projections
#pragma unroll
            for (int i = 0; i < kVRows; i++) {
                const int row = (tid >> 5) + 8 * i, col = tid & 31;
                const int64_t qi = q0 + row;
                double v = 0.0;
                if (qi < nq && k0 + col < d) {
                    v = static_cast<double>(q[qi * d + k0 + col]);
                    if (!(fabs(v) <= 1.79769313486231570815e+308)) badrow[row] = 1;
                }
                Vs[row][col] = static_cast<float>(v);
                nrm[i] += v * v;
            }
#pragma unroll
            for (int i = 0; i < kAVecs; i++) {
                const int e = (tid + i * 256) * 4;
                const int kk = e / kMfmaTileP, pc = e % kMfmaTileP;
                fsp_f4 a = {0.0f, 0.0f, 0.0f, 0.0f};
                if (k0 + kk < d) {
                    const float* src = alphaT32 + static_cast<int64_t>(k0 + kk) * P + p0 + pc;
                    if (a_vec && p0 + pc + 3 < P) a = *reinterpret_cast<const fsp_f4*>(src);
                    else {
#pragma unroll
                        for (int u = 0; u < 4; u++) if (p0 + pc + u < P) a[u] = src[u];
                    }
                }
                *reinterpret_cast<fsp_f4*>(&As[kk][pc]) = a;
            }
            __syncthreads();
            mfma_tile();
            __syncthreads();
        }
    }
    // row norms: reduce the 32 lanes that share a row, one writer per row
#pragma unroll
    for (int i = 0; i < kVRows; i++) {
        double s = nrm[i];
        for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if ((tid & 31) == 0) rnorm2[(tid >> 5) + 8 * i] = s;
    }
    __syncthreads();
    if (bad && blockIdx.y == 0 && tid < kMfmaTileQ && q0 + tid < nq) bad[q0 + tid] = badrow[tid];

    // per-row guard band (y units): gamma * ||v||_2 + f32-underflow floor
    const double gamma = 1.01 * static_cast<double>(d + 4) * 5.9604644775390625e-08 * alpha_norm_max;  // 2^-24
    if (tid < kMfmaTileQ) rnorm2[tid] = gamma * sqrt(rnorm2[tid]) * 1.0000001 + static_cast<double>(d) * 1.2e-38;
    __syncthreads();
    const int half = lane >> 5, c32 = lane & 31;
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int pb = p0 + wave * (32 * CT) + ct * 32;     // first projection of this 32-column tile (wave-uniform)
        const int p = pb + c32;
        const double rr = (p < P) ? r[p] : 0.0, ww = (p < P) ? omega[p] : 1.0;
        // bucket guess through the reciprocal (a fp64 divide per pair was as long as the block's whole MFMA phase): ANY guess is
        // fine, because a pair is accepted only when its error interval lies strictly inside the guessed bucket's own edges
        const double iw = 1.0 / ww;
        // the (t,d) tables whose projections fall into this tile: lane (half, seg) packs the bits of table td_first + seg for row-half `half`
        const int td_first = pb / m;
        const int td_last = min(P - 1, pb + 31) / m;
        const int nseg = (pb < P) ? td_last - td_first + 1 : 0;
        const int my_td = td_first + c32;
        const bool seg_on = c32 < nseg;
        const int c_lo = seg_on ? max(0, my_td * m - pb) : 0;
        const int c_hi = seg_on ? min(min(31, P - 1 - pb), my_td * m + m - 1 - pb) : -1;
        const int flen = c_hi - c_lo + 1;
        const int j_lo = pb + c_lo - my_td * m;
        const unsigned fmask = (flen >= 32) ? 0xFFFFFFFFu : ((1u << max(flen, 0)) - 1u);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                const int64_t qi = q0 + row;
                const bool live = (qi < nq) && (p < P);
                const double y = static_cast<double>(acc[rt][ct][reg]);
                const double band = rnorm2[row];
                const double fl = floor((y + rr) * iw);
                // bucket edges in y units; safe iff [y - band, y + band] lies strictly inside (edge_lo, edge_hi)
                // with a margin for the fp64 roundings of the edges and of the reference's own (y + r)/omega
                const double e_lo = fl * ww - rr, e_hi = (fl + 1.0) * ww - rr;
                const double mg = 8.9e-16 * (fabs(fl * ww) + fabs(ww) + fabs(rr) + fabs(y));
                const bool safe = live && (y - band > e_lo + mg) && (y + band < e_hi - mg) && (fabs(fl) < 2147483000.0) && !badrow[row];
                const int32_t h = java_d2i(fl);
                if (live && hashes) hashes[qi * P + p] = h;
                if (live && !safe) {
                    const unsigned long long pos = atomicAdd(fix_count, 1ull);
                    if (pos < static_cast<unsigned long long>(fix_cap)) fix_list[pos] = qi * P + p;
                }
                // Coding.C: bit pos = (lambda-1-i)*m + j  <-  bit i of (h_j ^ 0x80000000); one ballot per plane, low half = the
                // 32 columns of row `row` (half 0), high half = those of row + 4 (half 1)
                const uint32_t hj = static_cast<uint32_t>(h) ^ 0x80000000u;
                unsigned long long w0 = 0, w1 = 0, w2 = 0;
                for (int i = 0; i < lambda; i++) {                            // uniform trip count: every lane takes part in the ballot
                    const unsigned long long bm = __ballot(safe && ((hj >> (i & 31)) & 1u));
                    if (flen > 0) {
                        const unsigned long long field = (bm >> (half * 32 + c_lo)) & fmask;
                        const int bp = (lambda - 1 - i) * m + j_lo;
                        const int wd = bp >> 6, off = bp & 63;
                        const unsigned long long lo = field << off;
                        const unsigned long long hi = (off + flen > 64) ? (field >> (64 - off)) : 0ull;
                        if (wd == 0) { w0 |= lo; w1 |= hi; } else if (wd == 1) { w1 |= lo; w2 |= hi; } else { w2 |= lo; }
                    }
                }
                if (flen > 0 && qi < nq) {
                    unsigned long long* cw = codes + (qi * TD + my_td) * W;
                    if (w0) atomicOr(cw, w0);
                    if (W > 1 && w1) atomicOr(cw + 1, w1);
                    if (W > 2 && w2) atomicOr(cw + 2, w2);
                }
            }
        }
    }
}

// Exact re-computation of the flagged (query, projection) pairs — the reference's chain, acc = acc + v[k] * alpha[k] for k = 0 .. d - 1
// in that order (Coding.java:349-353) — and the pair's code bits OR-ed into the code words (the MFMA epilogue left them clear).
// A wave takes kFixG pairs at a time: for every 64 dimensions, the 64 lanes compute one pair's 64 PRODUCTS at once (the query row and
// the projection's row of alpha are both contiguous: two coalesced loads per pair and step) and park them in LDS; then lane g adds
// pair g's 64 products to its running sum IN ORDER.  The products are the same fp64 roundings whoever computes them, the additions
// keep their order: bit-identical to one lane walking the pair alone — which is what this kernel did, one lane per pair and two
// scattered loads per dimension: 220 us for a few hundred pairs at d = 768 / 960 whatever the batch (the MFMA path's fixed cost).
constexpr int kFixG = 8;            // pairs per wave task
constexpr int kFixCH = 4;           // 64-dimension steps per chunk: 8 x 4 x 2 = 64 loads in flight per lane and round trip to memory
template <typename TIn>
__global__ __launch_bounds__(256) void encode_fix_kernel(const TIn* __restrict__ q, int d, const double* __restrict__ alpha_rows /*[P][d]*/,
                                                         const double* __restrict__ r, const double* __restrict__ omega, int P, int m, int lambda, int W, int TD,
                                                         const int64_t* __restrict__ fix_list, const unsigned long long* __restrict__ fix_count,
                                                         int64_t fix_cap, int32_t* __restrict__ hashes, unsigned long long* __restrict__ codes) {
    constexpr int kChunk = 64 * kFixCH;
    __shared__ double prods[256 / 64][kFixG][kChunk + 1];      // (+1: lane g reads row g — rows a multiple of 64 doubles apart would share a bank)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long n = min(*fix_count, static_cast<unsigned long long>(fix_cap));
    const unsigned long long ntasks = (n + kFixG - 1) / kFixG;
    const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * (blockDim.x >> 6);
    for (unsigned long long task = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x >> 6) + wave; task < ntasks; task += nwaves) {   // wave-uniform
        const unsigned long long i = task * kFixG + lane;
        const bool mine = lane < kFixG && i < n;
        const int64_t e = mine ? fix_list[i] : -1;
        const int64_t qi = mine ? e / P : 0;
        const int p = mine ? static_cast<int>(e - qi * P) : 0;
        // every pair's row bases once per task (a pair past the end of the list re-reads pair 0's rows: computed, never used)
        const TIn* qrow[kFixG];
        const double* arow[kFixG];
#pragma unroll
        for (int g = 0; g < kFixG; g++) {
            qrow[g] = q + __shfl(qi, g) * d;
            arow[g] = alpha_rows + static_cast<int64_t>(__shfl(p, g)) * d;
        }
        double acc = 0.0;
        for (int k0 = 0; k0 < d; k0 += kChunk) {
            double pr[kFixG][kFixCH];
#pragma unroll
            for (int g = 0; g < kFixG; g++)
#pragma unroll
                for (int c = 0; c < kFixCH; c++) {
                    const int k = min(k0 + c * 64 + lane, d - 1);           // (clamped: a product past d is never added)
                    pr[g][c] = static_cast<double>(qrow[g][k]) * arow[g][k];
                }
#pragma unroll
            for (int g = 0; g < kFixG; g++)
#pragma unroll
                for (int c = 0; c < kFixCH; c++) prods[wave][g][c * 64 + lane] = pr[g][c];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (mine) {
                const int kn = min(kChunk, d - k0);
                const double* mine_row = prods[wave][lane];
                int j = 0;
                for (; j + 8 <= kn; j += 8) {                // eight LDS reads requested together, then added IN ORDER (Coding.java:351): one
                    double t[8];                              //   read latency per eight terms instead of one per term
#pragma unroll
                    for (int u = 0; u < 8; u++) t[u] = mine_row[j + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) acc = acc + t[u];
                }
                for (; j < kn; j++) acc = acc + mine_row[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (mine) {
            const double y = acc + r[p];
            const int32_t h = java_d2i(floor(y / omega[p]));
            if (hashes) hashes[e] = h;
            const uint32_t hj = static_cast<uint32_t>(h) ^ 0x80000000u;
            const int td = p / m, j = p - td * m;
            unsigned long long* cw = codes + (qi * TD + td) * W;
            for (int b = 0; b < lambda; b++)
                if ((hj >> (b & 31)) & 1u) {
                    const int pos = (lambda - 1 - b) * m + j;
                    atomicOr(cw + (pos >> 6), 1ull << (pos & 63));
                }
        }
    }
}

}  // namespace fspann
