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

template <typename TIn, int QB>
__global__ __launch_bounds__(kEncThreads) void encode_exact_kernel(
    const TIn* __restrict__ q, int64_t nq, int d, const double* __restrict__ alphaT,
    const double* __restrict__ r, const double* __restrict__ omega, int P, int m, int lambda, int W, int TD,
    int tdPerBlock, uint64_t* __restrict__ codes, int32_t* __restrict__ hashes, int32_t* __restrict__ bad,
    double* __restrict__ proj) {
    __shared__ double vs[QB * kEncDC];
    __shared__ int32_t Hs[QB * kEncThreads];
    __shared__ int badq[QB];

    const int tid = threadIdx.x;
    const int64_t q0 = static_cast<int64_t>(blockIdx.x) * QB;
    const int td0 = blockIdx.y * tdPerBlock;
    const int tdn = min(tdPerBlock, TD - td0);
    const int nproj = tdn * m;
    const bool active = tid < nproj;
    const int p = td0 * m + tid;

    if (tid < QB) badq[tid] = 0;

    double acc[QB];
#pragma unroll
    for (int i = 0; i < QB; i++) acc[i] = 0.0;

    for (int c0 = 0; c0 < d; c0 += kEncDC) {
        const int dc = min(kEncDC, d - c0);
        __syncthreads();
        for (int idx = tid; idx < QB * dc; idx += kEncThreads) {
            const int qq = idx / dc, i = idx - qq * dc;
            const int64_t qi = q0 + qq;
            double v = 0.0;
            if (qi < nq) {
                v = static_cast<double>(q[qi * d + c0 + i]);
                if (!(fabs(v) <= 1.79769313486231570815e+308)) atomicOr(&badq[qq], 1);  // NaN or Inf
            }
            vs[qq * kEncDC + i] = v;
        }
        __syncthreads();
        if (active) {
            const double* ap = alphaT + static_cast<int64_t>(c0) * P + p;
#pragma unroll 4
            for (int i = 0; i < dc; i++) {
                const double a = ap[static_cast<int64_t>(i) * P];
#pragma unroll
                for (int qq = 0; qq < QB; qq++) {
                    const double prod = vs[qq * kEncDC + i] * a;  // acc += a[i]*b[i], Coding.java:351
                    acc[qq] = acc[qq] + prod;
                }
            }
        }
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

    // Coding.C: bit pos = (lambda-1-i)*m + j  <-  bit i of (h_j ^ 0x80000000)
    const int bitsTotal = m * lambda;
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
    if (bad && blockIdx.y == 0 && tid < QB && q0 + tid < nq) bad[q0 + tid] = badq[tid];
}

}  // namespace fspann
