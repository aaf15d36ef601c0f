// tick.hip.h — one launch for three independent stages of three consecutive batches:
//
//     encode(batch t+2)   TokenGen coding            (encode_exact_block,   encode.hip.h)
//     route (batch t+1)   bounded select, probe fused (route_lazy_run,       route_lazy.hip.h)
//     refine(batch t)     L2 scan + top-K             (refine_scan_block,    refine.hip.h)
//
// Why: run one after the other, the three kernels leave the GPU mostly idle — Route is a ~30 us chain of dependent L2
// round trips per query with all queries resident at once (latency bound, HBM nearly idle), Refine streams 134 MB
// (bandwidth bound, the integer pipes idle), encode fills a quarter of the CUs.  The stages of ONE batch depend on each
// other, but a serving loop has batches in flight anyway (in production the host decrypts batch t's candidates while the
// GPU routes batch t+1), and stages of DIFFERENT batches are independent: each workgroup of this launch takes one role,
// roles are interleaved over the grid so every CU holds a mix, and the latency-bound workgroups run under the
// bandwidth-bound ones.  Nothing inside the launch waits for anything else in it (no spin, no grid sync): every
// workgroup runs to completion on its own, the dependencies are between LAUNCHES (stream order).
//
// A query the bounded select cannot hold (count = PENDING, probe lists in the hand-over buffer that travels with the
// batch) is redone with the full select by the workgroup that refines it one tick later — before it scans.  That removes
// the hand-back launch of fspann_route_dev (an empty dependent kernel costs ~4.5 us on this runtime).
#pragma once
#include "encode.hip.h"
#include "refine.hip.h"
#include "route_lazy.hip.h"

namespace fspann {

constexpr int kTickThreads = 256;
static_assert(kTickThreads == kLzThreads && kTickThreads == kRefRows && kTickThreads == kEncThreads, "one workgroup shape for all roles");
constexpr int kTickEncQB = 4;

// Each role's arguments are a kernel parameter of their own (one big by-value struct gets copied to scratch memory, and a
// kernel that needs scratch pays for its allocation at every dispatch).
struct TickHead {
    int n_enc, n_route, n_refine;   // workgroups per role (n_enc = enc_gx * enc_gy)
    int enc_gx;
    int route_front;                // the first route_front workgroups of the launch are route: the long jobs start first
    int has_fix;                    // `fix` is valid: PENDING queries of the batch being refined are redone before their scan
    int64_t nq_refine;              // queries of the batch being refined (n_refine workgroups share them)
    long long* dbg;                 // FSPANN_DEBUG_STAMPS builds: [grid][4] = {role, index, start, end} (wall_clock64), else unused
};
// route = bounded select of the batch being routed; fix_dev = Route of the batch being REFINED (kLds = false arenas, one
// slice per refine workgroup), in device memory; ref = the scan.

enum { kTickEncode = 0, kTickRoute = 1, kTickRefine = 2 };

// Role and index-within-role of workgroup b.  Dispatch order = grid order, so the order is the schedule: longest jobs
// first.  The encode workgroups head the grid (each walks the whole alpha matrix through ~16 dependent load rounds: 12 us
// on an idle memory system, 45 us measured when they start among bandwidth-bound neighbours), then a route-only stretch
// (30 us chains), then the rest of the route workgroups spread evenly between the refine ones (22 us, bandwidth bound), so
// that the tail of the launch consists of the short jobs.  (Bresenham line: exact counts, no table.)
__device__ __forceinline__ void tick_role(const TickHead& p, const int b, int* role, int* idx) {
    if (b < p.n_enc) { *role = kTickEncode; *idx = b; return; }
    const int b1 = b - p.n_enc;
    if (b1 < p.route_front) { *role = kTickRoute; *idx = b1; return; }
    const long long j = b1 - p.route_front;
    const long long rr = p.n_route - p.route_front;
    const long long M = rr + p.n_refine;
    const long long r0 = (j * rr) / M, r1 = ((j + 1) * rr) / M;
    if (r1 > r0) { *role = kTickRoute; *idx = p.route_front + static_cast<int>(r0); return; }
    *role = kTickRefine;
    *idx = static_cast<int>(j - r0);
}

template <bool GATHER>
__global__ __launch_bounds__(kTickThreads, 4) void tick_kernel(const TickHead h, const EncodeArgs<float> enc, const RouteParams route,
                                                               const RouteParams* __restrict__ fix_dev, const RefineArgs<float, float> ref) {
    extern __shared__ __align__(16) unsigned char smem[];
    int role, idx;
    tick_role(h, static_cast<int>(blockIdx.x), &role, &idx);    // block-uniform
#ifdef FSPANN_DEBUG_STAMPS
    if (h.dbg && threadIdx.x == 0) { h.dbg[blockIdx.x * 4 + 0] = role; h.dbg[blockIdx.x * 4 + 1] = idx; h.dbg[blockIdx.x * 4 + 2] = wall_clock64(); }
#endif
    if (role == kTickRoute) {
#ifndef TICK_NO_ROUTE
        route_lazy_run<kLzThreads>(route, smem, idx, h.n_route, idx);
#endif
    } else if (role == kTickRefine) {
        // One workgroup per query.  (A quarter of the workgroups streaming several queries each was 1.4-1.7x slower here than in
        // its own kernel while the scan still needed 152 registers — this kernel has 128 — and made the launch slower overall.)
        const int64_t qi = idx;                                   // nchunks == 1 (host): one workgroup per query
        int cnt = ref.cand_count[qi];
#ifndef TICK_NO_FIX
        if (h.has_fix && cnt == kRoutePending) {
            // the bounded select handed this query over one tick ago: finish its Route with the full select, then scan.
            // The redo's parameters are read from device memory HERE, in the rare path: as a second RouteParams kernel
            // argument they would sit in ~80 SGPRs for every role
            const RouteParams fix = *fix_dev;
            const int TP = fix.TD * fix.P;
            route_select_query<false, kTickThreads>(fix, smem, idx, fix.probe_g + qi * TP, fix.nprobe_g + qi * fix.TD, qi);
            __threadfence();
            __syncthreads();                                      // F_q and its count are in global memory, LDS is free again
            cnt = __hip_atomic_load(const_cast<int32_t*>(ref.cand_count) + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // not through a stale cache line
        }
#endif
        // dense blocks: the streaming scan's per-unit code (buffer loads, one tile in flight, nt policy) with one unit per
        // workgroup; the store gather keeps the scan block
        if constexpr (!GATHER) refine_stream_run<float, float, 32, false>(ref, smem, qi, static_cast<int64_t>(h.n_refine), h.nq_refine, true);
        else refine_scan_block<float, float, 32, true, GATHER>(ref, smem, idx, cnt);
    } else {
#ifndef TICK_NO_ENC
        encode_exact_block<float, kTickEncQB>(enc, idx % h.enc_gx, idx / h.enc_gx, reinterpret_cast<int32_t*>(smem));
#endif
    }
#ifdef FSPANN_DEBUG_STAMPS
    __syncthreads();
    if (h.dbg && threadIdx.x == 0) h.dbg[blockIdx.x * 4 + 3] = wall_clock64();
#endif
}

// encode(batch t+1) + Route(batch t) as one launch, nothing else: the two stages in front of the host's decrypt loop.  Without a
// Refine role the launch needs only the bounded select's LDS and registers — in its 512-entry class 19.6 KB and 80 registers, six
// workgroups per CU — and the encode workgroups (one wave per SIMD, bound by the issue rate of a lone wave) fill slots beside
// the Route workgroups instead of holding the chip for a launch of their own.
template <int kEnt, bool kChk, int kTD = 0, int kP = 0>
__global__ __launch_bounds__(kTickThreads, (kEnt <= 512 ? 6 : 4)) void front_kernel(const TickHead h, const EncodeArgs<float> enc, const RouteParams route) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int b = static_cast<int>(blockIdx.x);
    if (b < h.n_enc) encode_exact_block<float, kTickEncQB>(enc, b % h.enc_gx, b / h.enc_gx, reinterpret_cast<int32_t*>(smem));
    else route_lazy_run<kLzThreads, kEnt, kChk, kTD, kP>(route, smem, b - h.n_enc, h.n_route, b - h.n_enc);
}

// The stand-alone streaming scan (refine_stream_kernel) for a batch whose Route ran with a hand-over buffer: every workgroup
// finishes the Route of the PENDING queries among ITS units with the full select (normally none: one count load per unit,
// looked at while the first tile is already under way), then streams.  One 256-row chunk per query only (host: nchunks == 1), so a query's F_q is read by the workgroup that
// completed it.  This is what removes the hand-back launch from a serving loop that runs Route and Refine as separate launches.
struct RefineRouteFix {
    static constexpr bool enabled = true;
    const RouteParams* fix_dev;
    unsigned char* smem;
    __device__ __forceinline__ void operator()(const int64_t qi) const {
        const RouteParams fix = *fix_dev;                           // read in the rare path only (see tick_kernel)
        const int TP = fix.TD * fix.P;
        route_select_query<false, kRefRows>(fix, smem, static_cast<int>(blockIdx.x), fix.probe_g + qi * TP, fix.nprobe_g + qi * fix.TD, qi);
        __threadfence();
        __syncthreads();                                            // F_q and its count are in global memory, LDS is free again
    }
};
template <bool GATHER>
__global__ __launch_bounds__(kRefRows, (GATHER ? 2 : 4)) void refine_stream_fix_kernel(const RefineArgs<float, float> a, const int64_t nq,
                                                                                       const RouteParams* __restrict__ fix_dev) {
    extern __shared__ __align__(16) unsigned char smem[];
    refine_stream_run<float, float, 32, GATHER, RefineRouteFix>(a, smem, static_cast<int64_t>(blockIdx.x), static_cast<int64_t>(gridDim.x), nq, true,
                                                                RefineRouteFix{fix_dev, smem});
}

// The same redo as its own launch (fspann_tick_dev when the three roles cannot share one kernel): a workgroup per query,
// all but the PENDING ones leave at once.
__global__ __launch_bounds__(kTickThreads, 4) void tick_fix_kernel(RouteParams fix) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t qi = blockIdx.x;
    if (fix.out_count[qi] != kRoutePending) return;
    const int TP = fix.TD * fix.P;
    route_select_query<false, kTickThreads>(fix, smem, static_cast<int>(blockIdx.x), fix.probe_g + qi * TP, fix.nprobe_g + qi * fix.TD, qi);
}

}  // namespace fspann
