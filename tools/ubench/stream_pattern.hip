// Microbenchmark: read bandwidth of the refine kernel's access pattern (8 rows x 128 B per wave-instruction,
// rows 512 B apart, 4 passes) vs fully contiguous 1 KiB-per-wave-instruction reads, 134 MB buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
// pattern A: block = 256 rows x 128 floats; thread handles (row = wave*64 + v/8, cv = v%8) like refine_scan_kernel
__global__ __launch_bounds__(256) void patA(const float* __restrict__ x, float* out, int depth) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* base = x + (size_t)blockIdx.x * 256 * 128;
    f4 acc = {0, 0, 0, 0};
    for (int c0 = 0; c0 < 128; c0 += 32) {
        f4 r[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int v = lane + i * 64;
            const int row = wave * 64 + v / 8, cv = v % 8;
            r[i] = *reinterpret_cast<const f4*>(base + (size_t)row * 128 + c0 + cv * 4);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) acc += r[i];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1;
}
// pattern B: same bytes per block, contiguous: thread t reads f4 at base + (t + i*256)
__global__ __launch_bounds__(256) void patB(const float* __restrict__ x, float* out, int depth) {
    const f4* base = reinterpret_cast<const f4*>(x + (size_t)blockIdx.x * 256 * 128);
    f4 acc = {0, 0, 0, 0};
    for (int c0 = 0; c0 < 4; c0++) {
        f4 r[8];
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = base[threadIdx.x + (c0 * 8 + i) * 256];
#pragma unroll
        for (int i = 0; i < 8; i++) acc += r[i];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1;
}
// pattern C: all 32 loads issued up front (max memory-level parallelism), refine's addressing
__global__ __launch_bounds__(256) void patC(const float* __restrict__ x, float* out, int depth) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* base = x + (size_t)blockIdx.x * 256 * 128;
    f4 r[32];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int v = lane + i * 64;
            const int row = wave * 64 + v / 8, cv = v % 8;
            r[c * 8 + i] = *reinterpret_cast<const f4*>(base + (size_t)row * 128 + c * 32 + cv * 4);
        }
    f4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 32; i++) acc += r[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1;
}
int main() {
    const size_t n = (size_t)1024 * 256 * 128;
    float *x, *out;
    hipMalloc(&x, n * 4); hipMalloc(&out, 64);
    hipMemset(x, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, void (*k)(const float*, float*, int)) {
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, x, out, 0);
        hipDeviceSynchronize();
        float best = 1e9, sum = 0;
        for (int i = 0; i < 20; i++) {
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, x, out, 0); hipEventRecord(e1);
            hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
        }
        printf("%-28s avg %.1f us  min %.1f us  -> %.0f GB/s\n", name, sum / 20 * 1e3, best * 1e3, n * 4 / (sum / 20) / 1e6);
    };
    run("A refine pattern (4 passes)", patA);
    run("B contiguous (4 passes)", patB);
    run("C refine pattern, 32 in flight", patC);
    return 0;
}
