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
__global__ __launch_bounds__(256, WGS_PER_CU) void pat(const char* __restrict__ x, unsigned* out, int nunits) {
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
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) out[0] = 1;
}

int main(int argc, char** argv) {
    const int nunits = 1024;
    const size_t unit_set = (size_t)nunits * 131072;      // 128 MiB per launch
    const int nsets = 32;                                 // 4 GiB cycled
    char* x; unsigned* out;
    if (hipMalloc(&x, unit_set * nsets) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(x, 0x5A, unit_set * nsets);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int grid) {
        std::vector<float> ts;
        for (int it = 0; it < 8 + 48; it++) {
            const char* p = x + (size_t)(it % nsets) * unit_set;
            hipDeviceSynchronize();
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, e0, e1, 0, p, out, nunits);
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
