// build.hip.h — Setup on the GPU: GreedyPartitioner.build (idx/GreedyPartitioner.java:37-76) for all T*D tables.
//
// The reference, per (t,d): iterate HashMap<String,BitSet>(staged.size()) (PIS:413-420), compute the 63-bit key of every
// code, List.sort by key (stable), cut blocks of 64, take first / last key and the middle element's code per block.  So
// elements are ordered by (key, HashMap iteration position), and the iteration position of an id is (bin at the map's
// final table length, insertion order inside the bin) while no bin is treeified (checked on the host before this runs).
//
// Here: the (bin, insertion position) order is the same for every table, so it is established ONCE — a stable LSD radix
// sort of the staged positions by bin — and each table then only needs a STABLE sort of that sequence by key: an LSD radix
// sort over the key's significant bytes (keys carry the first 63 code bits MSB-first, so only bits [63 - bits, 62] vary).
// One pass = digit histogram per tile, exclusive scan, stable scatter (ranks from wave ballots: the order inside a wave
// round is lane order, rounds and waves of a tile are laid out in input order).  The cut is then element-parallel.
// Integer / byte work, HBM bound: nothing here is a GEMM.
#pragma once
#include "fspann_common.h"

namespace fspann {

constexpr int kRsThreads = 256;
constexpr int kRsRounds = 8;                                   // 64-item rounds per wave
constexpr int kRsTile = kRsThreads * kRsRounds;                // items per workgroup
constexpr int kRsWaves = kRsThreads / 64;

__global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift, uint32_t* __restrict__ hist,
                                                             int nblocks, uint32_t* __restrict__ tot) {
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kRsTile;
    for (int r = 0; r < kRsRounds; r++) {
        const int64_t i = base + r * kRsThreads + tid;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[static_cast<int64_t>(tid) * nblocks + blockIdx.x] = h[tid];
    if (h[tid]) atomicAdd(&tot[tid], h[tid]);      // digit totals of this pass: the scan's workgroups start from their prefix
}

// hist[d][b] -> exclusive offset of (digit d, block b) in the output: all smaller digits first, then earlier blocks.
// One workgroup PER DIGIT (a single 256-thread workgroup walking every row serially took 204 us per pass at n = 1 M — 17 ms of a
// 160 ms Setup): its base is the prefix of the digit totals rs_hist_kernel accumulated, its row is scanned 256 blocks at a time.
// tot_next = the totals of the NEXT pass, cleared here (the two arrays alternate).
__global__ __launch_bounds__(256) void rs_scan_kernel(uint32_t* __restrict__ hist, int nblocks, const uint32_t* __restrict__ tot_cur,
                                                      uint32_t* __restrict__ tot_next) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_base;
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (d == 0) tot_next[tid] = 0;
    uint32_t v = (tid < d) ? tot_cur[tid] : 0u;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) wsum[wave] = v;
    __syncthreads();
    if (tid == 0) s_base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    uint32_t carry = s_base;
    uint32_t* row = hist + static_cast<int64_t>(d) * nblocks;
    for (int b0 = 0; b0 < nblocks; b0 += 256) {
        const int b = b0 + tid;
        const uint32_t x = (b < nblocks) ? row[b] : 0u;
        uint32_t incl = x;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        __syncthreads();                       // wsum of the previous chunk has been read by everyone
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (b < nblocks) row[b] = carry + before + incl - x;
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pay, int64_t n, int shift,
                                                                const uint32_t* __restrict__ offs, int nblocks, uint64_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ pay_out) {
    __shared__ uint32_t wcount[kRsWaves][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kRsWaves * 256; i += kRsThreads) (&wcount[0][0])[i] = 0;
    __syncthreads();
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kRsTile + static_cast<int64_t>(wave) * (kRsRounds * 64);
    uint64_t kreg[kRsRounds];
    uint32_t preg[kRsRounds], rank[kRsRounds];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) {
        const int64_t i = base + r * 64 + lane;
        const bool valid = i < n;
        kreg[r] = valid ? keys[i] : 0ull;
        preg[r] = valid ? pay[i] : 0u;
        const unsigned d = static_cast<unsigned>(kreg[r] >> shift) & 255u;
        unsigned long long mask = __ballot(valid);            // lanes of this round holding the same digit
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool b = (d >> bit) & 1u;
            const unsigned long long bm = __ballot(valid && b);
            mask &= b ? bm : ~bm;
        }
        uint32_t prior = 0;
        if (valid) prior = wcount[wave][d];                   // every lane reads before the group's first lane writes (program order)
        rank[r] = prior + static_cast<uint32_t>(__popcll(mask & lt));
        if (valid && (mask & lt) == 0ull) wcount[wave][d] = prior + static_cast<uint32_t>(__popcll(mask));
    }
    __syncthreads();
    {   // digit `tid`: where each wave's share starts in the output
        uint32_t run = offs[static_cast<int64_t>(tid) * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < kRsWaves; w++) {
            const uint32_t t = wcount[w][tid];
            wcount[w][tid] = run;
            run += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) {
        const int64_t i = base + r * 64 + lane;
        if (i < n) {
            const unsigned d = static_cast<unsigned>(kreg[r] >> shift) & 255u;
            const uint32_t dst = wcount[wave][d] + rank[r];
            keys_out[dst] = kreg[r];
            pay_out[dst] = preg[r];
        }
    }
}

// keys[i] = bin of staged position i, pay[i] = i
__global__ void build_bin_keys_kernel(const uint32_t* __restrict__ bucket, int64_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ pay) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = bucket[i]; pay[i] = static_cast<uint32_t>(i); }
}
// For table td: element i of the (bin, position)-ordered sequence -> its 63-bit key (GreedyPartitioner.computeKey :87-96:
// code bit b -> key bit 62 - b for b < 63) and its staged position.
__global__ void build_table_keys_kernel(const uint64_t* __restrict__ codes, int TD, int W, int td, const int32_t* __restrict__ ord,
                                        const uint32_t* __restrict__ perm0, int64_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ pay) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pos = perm0[i];
    const int64_t h = ord[pos];
    keys[i] = __brevll(codes[(h * TD + td) * W]) >> 1;
    pay[i] = pos;
}
// The cut: one thread per element.  ids[i] = handle of sorted element i; the first element of block p writes the block's
// min / max key, id offset and representative (code of the middle element, mid = i0 + ((i1 - i0 - 1) >>> 1)).
__global__ void build_cut_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pay, const int32_t* __restrict__ ord,
                                 const uint64_t* __restrict__ codes, int TD, int W, int td, int64_t n, int S, int64_t* __restrict__ min_key,
                                 int64_t* __restrict__ max_key, uint64_t* __restrict__ rep, int64_t* __restrict__ off, int32_t* __restrict__ ids) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ids[i] = ord[pay[i]];
    if (i % S == 0) {
        const int64_t p = i / S, i1 = min(i + S, n);
        min_key[p] = static_cast<int64_t>(keys[i]);
        max_key[p] = static_cast<int64_t>(keys[i1 - 1]);
        off[p] = i;
        const int64_t mid = i + ((i1 - i - 1) >> 1);
        const int64_t hm = ord[pay[mid]];
        for (int w = 0; w < W; w++) rep[p * W + w] = codes[(hm * TD + td) * W + w];
        if (i1 == n) off[p + 1] = n;
    }
}

}  // namespace fspann
