// Microbenchmark (dev tool, GPU box): what does the ADDRESS PATTERN of the refinement scan cost on cold HBM?
// One launch = 1024 workgroups x 256 threads, each reading its own 128 KB unit (256 rows x 512 B) with 16-byte nt loads;
// 32 units-sets (4 GiB) are cycled so no launch finds its data in the 256 MiB Infinity Cache.  Variants:
//   seg128  the scan's pattern: 4 passes, a wave-instruction = 8 rows x 128 B (rows 512 B apart), 2 passes in flight
//   seg256  2 passes, a wave-instruction = 4 rows x 256 B, 1 pass (16 loads) ahead
//   rows    1 pass, a wave-instruction = 2 whole rows (1 KB contiguous), a wave sweeps its 32 KB in order, 8 loads per step, 2 steps in flight
//   wgseq   as rows, but the WORKGROUP sweeps its 128 KB in order (instruction i of all waves = 4 KB contiguous)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/scan_pattern tools/ubench/scan_pattern.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define LD(p) __builtin_nontemporal_load(reinterpret_cast<const u4*>(p))

template <int MODE, int WGS_PER_CU>
__global__ __launch_bounds__(256, WGS_PER_CU) void pat(const char* __restrict__ x, unsigned* out, int nunits, const float* __restrict__ qv) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u4 acc = {0, 0, 0, 0};
    for (int u = blockIdx.x; u < nunits; u += gridDim.x) {
        const char* base = x + (size_t)u * 131072;
        if (MODE == 0) {            // seg128: 4 passes x 8 loads, two passes in flight
            u4 a[8], b[8];
            auto addr = [&](int c, int i) { const int v = lane + i * 64; return base + (size_t)(wave * 64 + v / 8) * 512 + c * 128 + (v % 8) * 16; };
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = LD(addr(0, i));
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = LD(addr(1, i));
#pragma unroll
            for (int i = 0; i < 8; i++) { acc ^= a[i]; a[i] = LD(addr(2, i)); }
#pragma unroll
            for (int i = 0; i < 8; i++) { acc ^= b[i]; b[i] = LD(addr(3, i)); }
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= a[i];
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= b[i];
        } else if (MODE == 1) {     // seg256: 2 passes x 16 loads, both requested up front
            u4 a[16], b[16];
            auto addr = [&](int c, int i) { const int v = lane + i * 64; return base + (size_t)(wave * 64 + v / 16) * 512 + c * 256 + (v % 16) * 16; };
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = LD(addr(0, i));
#pragma unroll
            for (int i = 0; i < 16; i++) b[i] = LD(addr(1, i));
#pragma unroll
            for (int i = 0; i < 16; i++) acc ^= a[i];
#pragma unroll
            for (int i = 0; i < 16; i++) acc ^= b[i];
        } else if (MODE == 2) {     // rows: the wave sweeps its 32 KB in order, 4 steps x 8 loads, two steps in flight
            u4 a[8], b[8];
            auto addr = [&](int c, int i) { return base + (size_t)wave * 32768 + (size_t)(c * 8 + i) * 1024 + lane * 16; };
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = LD(addr(0, i));
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = LD(addr(1, i));
#pragma unroll
            for (int i = 0; i < 8; i++) { acc ^= a[i]; a[i] = LD(addr(2, i)); }
#pragma unroll
            for (int i = 0; i < 8; i++) { acc ^= b[i]; b[i] = LD(addr(3, i)); }
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= a[i];
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= b[i];
        } else if (MODE == 3) {     // wgseq: the workgroup sweeps its 128 KB in order
            u4 a[8], b[8];
            auto addr = [&](int c, int i) { return base + (size_t)(c * 8 + i) * 4096 + tid * 16; };
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = LD(addr(0, i));
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = LD(addr(1, i));
#pragma unroll
            for (int i = 0; i < 8; i++) { acc ^= a[i]; a[i] = LD(addr(2, i)); }
#pragma unroll
            for (int i = 0; i < 8; i++) { acc ^= b[i]; b[i] = LD(addr(3, i)); }
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= a[i];
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= b[i];
        } else if (MODE == 4) {     // seg128, ONE pass in flight (half the outstanding bytes)
            auto addr = [&](int c, int i) { const int v = lane + i * 64; return base + (size_t)(wave * 64 + v / 8) * 512 + c * 128 + (v % 8) * 16; };
            for (int c = 0; c < 4; c++) {
                u4 a[8];
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = LD(addr(c, i));
#pragma unroll
                for (int i = 0; i < 8; i++) acc ^= a[i];
            }
        } else if (MODE == 5) {     // rows, all 32 loads of the wave's 32 KB up front
            u4 a[32];
#pragma unroll
            for (int i = 0; i < 32; i++) a[i] = LD(base + (size_t)wave * 32768 + (size_t)i * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < 32; i++) acc ^= a[i];
        } else if (MODE == 9 || MODE == 10) {   // seg64: a wave-instruction = 16 rows x 64 B; 8 passes of 4 loads, MODE 9: one pass in flight, MODE 10: two
            auto addr = [&](int c, int i) { const int v = lane + i * 64; return base + (size_t)(wave * 64 + v / 4) * 512 + c * 64 + (v % 4) * 16; };
            u4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = LD(addr(0, i));
            if (MODE == 10) {
#pragma unroll
                for (int i = 0; i < 4; i++) b[i] = LD(addr(1, i));
            }
            for (int c = 0; c < 8; c += 2) {
                if (MODE == 9) {
#pragma unroll
                    for (int i = 0; i < 4; i++) { b[i] = LD(addr(c + 1, i)); acc ^= a[i]; }
#pragma unroll
                    for (int i = 0; i < 4; i++) { if (c + 2 < 8) a[i] = LD(addr(c + 2, i)); acc ^= b[i]; }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) { acc ^= a[i]; if (c + 2 < 8) a[i] = LD(addr(c + 2, i)); }
#pragma unroll
                    for (int i = 0; i < 4; i++) { acc ^= b[i]; if (c + 3 < 8) b[i] = LD(addr(c + 3, i)); }
                }
            }
        } else if (MODE >= 6) {     // seg128 as MODE 0, every pass staged through LDS like the scan: 16-byte writes into a row-major
                                    // tile (pitch 144 B), wave sync, every lane reads ITS row back.  MODE 7: + the fp64 chain
                                    // s += (q - x)^2 (q wave-uniform).  MODE 8: + an epilogue of two workgroup barriers, an LDS
                                    // exchange and a few global stores.
            extern __shared__ __align__(16) unsigned char smem[];
            float* tile = reinterpret_cast<float*>(smem);
            constexpr int PITCH = 36;
            u4 a[8], b[8];
            // buffer loads like the scan: one resource per unit, a 32-bit lane offset + a scalar offset per slot, nt policy
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 131072, 0x00020000);
            const int slot_row = wave * 64 + lane / 8, slot_col = (lane % 8) * 4;
            const int slot_off = slot_row * 512 + slot_col * 4;
            auto bload = [&](int c, int i) { return __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs, slot_off + c * 128, i * 8 * 512, 2)); };
            double s = 0.0;
            typedef const float __attribute__((address_space(4)))* crow_t;
            const crow_t qrow = (crow_t)(qv + (u & 1023) * 128);
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define TILE(REG, C, NEXT)                                                                                     \
            do {                                                                                               \
                WSYNC();                                                                                       \
                _Pragma("unroll") for (int i = 0; i < 8; i++)                                                  \
                    *reinterpret_cast<u4*>(tile + (slot_row + i * 8) * PITCH + slot_col) = REG[i];             \
                if (NEXT >= 0) { _Pragma("unroll") for (int i = 0; i < 8; i++) REG[i] = bload(NEXT < 0 ? 0 : NEXT, i); } \
                WSYNC();                                                                                       \
                const float* myrow = tile + tid * PITCH;                                                       \
                _Pragma("unroll 4") for (int kk = 0; kk < 32; kk += 4) {                                       \
                    const u4 xv = *reinterpret_cast<const u4*>(myrow + kk);                                    \
                    if (MODE == 6) acc ^= xv;                                                                  \
                    else {                                                                                     \
                        _Pragma("unroll") for (int e = 0; e < 4; e++) {                                        \
                            const double q0 = static_cast<double>(qrow[(C) * 32 + kk + e]);                    \
                            const double x0 = static_cast<double>(__uint_as_float(xv[e]));                     \
                            const double d0 = q0 - x0;                                                         \
                            const double p0 = d0 * d0;                                                         \
                            s = s + p0;                                                                        \
                        }                                                                                      \
                    }                                                                                          \
                }                                                                                              \
            } while (0)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = bload(0, i);
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = bload(1, i);
            TILE(a, 0, 2);
            TILE(b, 1, 3);
            TILE(a, 2, -1);
            TILE(b, 3, -1);
            if (MODE >= 7) { acc.x ^= (unsigned)__double_as_longlong(sqrt(s)); acc.y ^= (unsigned)(__double_as_longlong(s) >> 32); }
            if (MODE >= 8) {
                __shared__ unsigned s_x[4];
                if (lane == 0) s_x[wave] = acc.x;
                __syncthreads();
                const unsigned m = s_x[0] ^ s_x[1] ^ s_x[2] ^ s_x[3];
                tile[tid] = __uint_as_float(acc.y ^ m);
                __syncthreads();
                if (tid < 10) { out[64 + u * 16 + tid] = __float_as_uint(tile[(tid * 7) & 255]); }
                __syncthreads();
            }
#undef TILE
#undef WSYNC
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) out[0] = 1;
}

int main(int argc, char** argv) {
    const int nunits = 1024;
    const size_t unit_set = (size_t)nunits * 131072;      // 128 MiB per launch
    const int nsets = 32;                                 // 4 GiB cycled
    char* x; unsigned* out;
    if (hipMalloc(&x, unit_set * nsets) != hipSuccess || hipMalloc(&out, 64 + 1024 * 64 + 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(x, 0x5A, unit_set * nsets);
    float* qv; hipMalloc(&qv, 1024 * 128 * 4); hipMemset(qv, 0x3C, 1024 * 128 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int grid) {
        std::vector<float> ts;
        for (int it = 0; it < 8 + 48; it++) {
            const char* p = x + (size_t)(it % nsets) * unit_set;
            hipDeviceSynchronize();
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), 256 * 36 * 4, 0, e0, e1, 0, p, out, nunits, qv);
            { hipError_t le = hipGetLastError(); if (le != hipSuccess && it == 0) printf("launch error: %s\n", hipGetErrorString(le)); }
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 8) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        float sum = 0; for (float t : ts) sum += t;
        const float avg = sum / ts.size(), med = ts[ts.size() / 2];
        printf("%-44s grid %4d  avg %6.2f us  med %6.2f us  min %6.2f us  -> %5.2f TB/s (avg)\n", name, grid, avg * 1e3, med * 1e3, ts[0] * 1e3, unit_set / (avg * 1e-3) / 1e12);
    };
    run("seg128 2 passes in flight (the scan) 4/CU", pat<0, 4>, 1024);
    run("seg128 + LDS staging 4/CU", pat<6, 4>, 1024);
    run("seg128 + LDS staging + fp64 chain 4/CU", pat<7, 4>, 1024);
    run("seg128 + LDS + fp64 + epilogue 4/CU", pat<8, 4>, 1024);
    run("seg64 (16 rows x 64 B per instr) 1 pass in flight 4/CU", pat<9, 4>, 1024);
    run("seg64 2 passes in flight 4/CU", pat<10, 4>, 1024);
    run("seg64 1 pass in flight, 6/CU (grid 1024)", pat<9, 6>, 1024);
    run("seg128 2 passes in flight (again)", pat<0, 4>, 1024);
    run("seg128 1 pass in flight 4/CU", pat<4, 4>, 1024);
    run("seg256 both passes up front 2/CU", pat<1, 2>, 1024);
    run("seg256 both passes up front 2/CU grid 512", pat<1, 2>, 512);
    run("rows (wave sweeps 32 KB) 2 steps in flight 4/CU", pat<2, 4>, 1024);
    run("rows, all 32 loads up front 2/CU", pat<5, 2>, 1024);
    run("rows, all 32 loads up front 2/CU grid 512", pat<5, 2>, 512);
    run("wgseq (WG sweeps 128 KB) 2 steps in flight 4/CU", pat<3, 4>, 1024);
    run("seg128 2 in flight, grid 768 (3/CU)", pat<0, 4>, 768);
    run("seg128 2 in flight, grid 512 (2/CU)", pat<0, 4>, 512);
    run("rows 2 in flight, grid 512 (2/CU)", pat<2, 4>, 512);
    run("wgseq 2 in flight, grid 512 (2/CU)", pat<3, 4>, 512);
    return 0;
}
