// groundtruth.hip.h — exact brute-force k-NN ground truth and the evaluation metrics.
//
// Restates api/src/main/java/com/fspann/api/GroundtruthPrecompute.java:142-163 (l2sq), :167-189 (HeapK, BY_D_THEN_ID) and
// :218-272 (run): for every query the k base vectors with the smallest SQUARED distance, ties by LOWER id, ids ascending by
// (distance, id).  The distance arithmetic is the reference's: per dimension `double d = q[i] - v` with q and v floats — a
// FLOAT subtraction (Java's binary numeric promotion), widened, then `sum += d * d` sequentially in fp64, no sqrt.  Each lane
// owns one base row and keeps kGtQT running sums (one per query of its tile), so every row element is loaded once per query
// tile and the sums are bit-identical to the JVM's whatever the launch shape.
// And ForwardSecureANNSystem.computeMetricsAtK (FSA:770-835): recall@k = |ann[0..k) ∩ gt[0..k)| / k; distance ratio@k =
// mean_i d(q, ann_i) / d(q, gt_i) over i < k with BaseVectorReader.l2 (FSA:1017-1073: `double d = q[i] - v` with q a
// double[], sqrt of the sequential sum), NaN unless all k terms exist and every d(q, gt_i) > 0.
//
// Selection is exact and HBM-shaped: the [Q x N] fp64 distances go to a scratch matrix; per query one workgroup finds the
// k-th smallest COMPOSITE key (distance bits, id) by a 12-pass MSB radix select over 8-bit digits (the composite is unique,
// so ties need no special case), then collects the k elements at or below it and orders them.
#pragma once
#include "fspann_common.h"

#pragma clang fp contract(off)

namespace fspann {

constexpr int kGtQT = 16;          // queries per tile of gt_dist_kernel
constexpr int kGtRows = 256;       // base rows (= lanes) per workgroup
constexpr int kGtSelThreads = 1024;
constexpr int kGtMaxK = 1024;

// dist[q][r] = sum_i (double)(q[q][i] - base[r][i])^2 for the queries [q0, q0 + kGtQT) of this tile.
__global__ __launch_bounds__(kGtRows) void gt_dist_kernel(const float* __restrict__ base, int64_t n, const float* __restrict__ q, int64_t nq, int d,
                                                          double* __restrict__ dist) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kGtRows + threadIdx.x;
    const int64_t q0 = static_cast<int64_t>(blockIdx.y) * kGtQT;
    typedef const float __attribute__((address_space(4)))* const_row_t;     // uniform loads -> scalar loads
    const_row_t qt[kGtQT];
#pragma unroll
    for (int t = 0; t < kGtQT; t++) qt[t] = (const_row_t)(q + min(q0 + t, nq - 1) * d);
    double acc[kGtQT];
#pragma unroll
    for (int t = 0; t < kGtQT; t++) acc[t] = 0.0;
    if (r < n) {
        const float* row = base + r * d;
        for (int i = 0; i < d; i++) {
            const float v = row[i];
#pragma unroll
            for (int t = 0; t < kGtQT; t++) {
                const float df = qt[t][i] - v;                    // float - float (GroundtruthPrecompute.java:150)
                const double dd = static_cast<double>(df);
                const double sq = dd * dd;
                acc[t] = acc[t] + sq;
            }
        }
#pragma unroll
        for (int t = 0; t < kGtQT; t++)
            if (q0 + t < nq) dist[(q0 + t) * n + r] = acc[t];
    }
}

// One workgroup per query: ids of the k smallest (distance, id), ascending.  keys = the query's row of `dist` (fp64 bits of
// non-negative values are monotone as uint64; a NaN distance sorts last, like Double.compare).
__global__ __launch_bounds__(kGtSelThreads) void gt_select_kernel(const double* __restrict__ dist, int64_t n, int k, int32_t* __restrict__ out_ids,
                                                                  double* __restrict__ out_d2) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long s_pk;
    __shared__ unsigned s_pi, s_need, s_cnt;
    __shared__ unsigned long long sel_key[kGtMaxK];
    __shared__ unsigned sel_id[kGtMaxK];
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(dist + qi * n);
    const int kk = static_cast<int>(min(static_cast<int64_t>(k), n));
    if (tid == 0) { s_pk = 0ull; s_pi = 0u; s_need = static_cast<unsigned>(kk); s_cnt = 0u; }
    __syncthreads();
    // 12 digits of 8 bits, most significant first: 8 from the distance bits, 4 from the id
    for (int p = 0; p < 12; p++) {
        for (int i = tid; i < 256; i += kGtSelThreads) hist[i] = 0u;
        __syncthreads();
        const unsigned long long pk = s_pk;
        const unsigned pi = s_pi;
        for (int64_t i = tid; i < n; i += kGtSelThreads) {
            const unsigned long long key = keys[i];
            unsigned digit;
            bool match;
            if (p < 8) {
                const int sh = 56 - 8 * p;
                match = (p == 0) || ((key >> (sh + 8)) == (pk >> (sh + 8)));
                digit = static_cast<unsigned>(key >> sh) & 255u;
            } else {
                const int sh = 24 - 8 * (p - 8);
                const unsigned id = static_cast<unsigned>(i);
                match = (key == pk) && ((p == 8) || ((id >> (sh + 8)) == (pi >> (sh + 8))));
                digit = (id >> sh) & 255u;
            }
            if (match) atomicAdd(&hist[digit], 1u);
        }
        __syncthreads();
        if (tid == 0) {       // the digit bucket holding the need-th smallest of the matching elements
            unsigned cum = 0, dsel = 255;
            const unsigned need = s_need;
            for (unsigned b = 0; b < 256; b++) {
                if (cum + hist[b] >= need) { dsel = b; break; }
                cum += hist[b];
            }
            s_need = need - cum;
            if (p < 8) s_pk |= static_cast<unsigned long long>(dsel) << (56 - 8 * p);
            else s_pi |= dsel << (24 - 8 * (p - 8));
        }
        __syncthreads();
    }
    // (s_pk, s_pi) is the kk-th smallest composite: collect everything at or below it (exactly kk elements), then order them
    const unsigned long long tk = s_pk;
    const unsigned ti = s_pi;
    for (int64_t i = tid; i < n; i += kGtSelThreads) {
        const unsigned long long key = keys[i];
        if (key < tk || (key == tk && static_cast<unsigned>(i) <= ti)) {
            const unsigned at = atomicAdd(&s_cnt, 1u);
            if (at < static_cast<unsigned>(kGtMaxK)) { sel_key[at] = key; sel_id[at] = static_cast<unsigned>(i); }
        }
    }
    __syncthreads();
    const int cnt = static_cast<int>(min(s_cnt, static_cast<unsigned>(kk)));
    for (int e = tid; e < cnt; e += kGtSelThreads) {
        const unsigned long long mk = sel_key[e];
        const unsigned mi = sel_id[e];
        int rank = 0;
        for (int j = 0; j < cnt; j++) rank += (sel_key[j] < mk) || (sel_key[j] == mk && sel_id[j] < mi);
        out_ids[qi * k + rank] = static_cast<int32_t>(mi);
        if (out_d2) out_d2[qi * k + rank] = __longlong_as_double(static_cast<long long>(mk));
    }
    for (int e = cnt + tid; e < k; e += kGtSelThreads) {
        out_ids[qi * k + e] = -1;
        if (out_d2) out_d2[qi * k + e] = __longlong_as_double(0x7FF0000000000000LL);
    }
}

// computeMetricsAtK (FSA:770-835) for one query per workgroup (64 lanes).  ann = [nq][ann_stride] ids (count per query),
// gt = [nq][gt_stride] ground-truth ids (>= k of them).  recall[q], ratio[q] (NaN when the reference returns NaN).
__global__ __launch_bounds__(64) void gt_metrics_kernel(const float* __restrict__ base, int64_t n, const float* __restrict__ q, int d, int k,
                                                        const int32_t* __restrict__ ann, int64_t ann_stride, const int32_t* __restrict__ ann_count,
                                                        const int32_t* __restrict__ gt, int64_t gt_stride, double* __restrict__ recall,
                                                        double* __restrict__ ratio) {
    const int64_t qi = blockIdx.x;
    const int lane = threadIdx.x;
    const int na = ann_count ? max(0, min(ann_count[qi], static_cast<int>(ann_stride))) : static_cast<int>(ann_stride);
    const int32_t* a = ann + qi * ann_stride;
    const int32_t* g = gt + qi * gt_stride;
    // recall: hits among the first min(k, na) results that are in gt[0..k)  (a Set: a repeated id counts each time it appears, like the Java loop)
    int hits = 0;
    for (int i = lane; i < min(k, na); i += 64) {
        const int32_t id = a[i];
        bool in = false;
        for (int j = 0; j < k; j++) in = in || (g[j] == id);
        hits += in ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) hits += __shfl_xor(hits, off);
    // ratio: needs k results; BaseVectorReader.l2 = sqrt(sum (q_i - v_i)^2), q widened to double first (FSA:1017-1073)
    // the reference adds the k terms in index order: rounds of 64 terms, lane l holds term 64 r + l, and a sequential fold over
    // the lanes inside every round reproduces the Java sum exactly (any k)
    double tot = 0.0;
    int usedt = 0;
    if (na >= k) {
        const float* qr = q + qi * d;
        for (int i0 = 0; i0 < k; i0 += 64) {
            const int i = i0 + lane;
            double term = 0.0;
            int used = 0;
            if (i < k) {
                const int32_t ai = a[i], gi = g[i];
                if (!(ai < 0 || ai >= n || gi < 0 || gi >= n)) {
                    double sg = 0.0, sa = 0.0;
                    for (int t = 0; t < d; t++) {
                        const double qv = static_cast<double>(qr[t]);
                        const double dg = qv - static_cast<double>(base[static_cast<int64_t>(gi) * d + t]);
                        const double pg = dg * dg;
                        sg = sg + pg;
                        const double da = qv - static_cast<double>(base[static_cast<int64_t>(ai) * d + t]);
                        const double pa = da * da;
                        sa = sa + pa;
                    }
                    const double dGt = sqrt(sg);
                    if (dGt > 0) { term = sqrt(sa) / dGt; used = 1; }
                }
            }
            for (int l = 0; l < 64; l++) {
                const double v = __shfl(term, l);
                const int u = __shfl(used, l);
                if (u) { tot = tot + v; usedt += u; }
            }
        }
    }
    if (lane == 0) {
        recall[qi] = static_cast<double>(hits) / static_cast<double>(k);
        ratio[qi] = (na >= k && usedt == k) ? tot / static_cast<double>(k) : __longlong_as_double(0x7FF8000000000000LL);
    }
}

}  // namespace fspann
