// Microbenchmark (dev tool): issue cost of the vector instructions the refinement scan spends its time in, one wave per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int OP> __global__ void k(const float* in, double* out, long long* cyc) {
    float x = in[threadIdx.x];
    double a = x, b = x * 0.5, c = 0.0, d0 = 1.0;
    float y = x;
    int s = 0;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(y)); }
            if (OP == 1) { asm volatile("v_add_f64 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b)); }
            if (OP == 2) { asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b)); }
            if (OP == 3) { asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(y)); }
            if (OP == 4) { asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(y), "v"(0x1f8) : "vcc"); }
            if (OP == 5) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(c) : "v"(b)); }   // dependent chain
            if (OP == 6) { asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d0) : "v"(a), "v"(b), "v"(c)); }
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[threadIdx.x] = a + c + d0 + s;
}
int main() {
    float* in; double* out; long long* cyc;
    hipMalloc(&in, 1024 * 4); hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 64 * 8);
    hipMemset(in, 0, 1024 * 4);
    const char* names[] = {"v_cvt_f64_f32", "v_add_f64", "v_mul_f64", "v_readlane_b32", "v_cmp_class_f32", "v_add_f64 (dependent)", "v_fma_f64"};
    for (int waves = 1; waves <= 4; waves *= 4) {
        for (int op = 0; op < 7; op++) {
            for (int rep = 0; rep < 2; rep++) {
                // one workgroup of `waves` x 4 waves: `waves` waves per SIMD
                dim3 g(1), b(64 * 4 * waves);
                switch (op) {
                    case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, in, out, cyc); break;
                    case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, in, out, cyc); break;
                    case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, in, out, cyc); break;
                    case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, in, out, cyc); break;
                    case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, in, out, cyc); break;
                    case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, in, out, cyc); break;
                    case 6: hipLaunchKernelGGL(k<6>, g, b, 0, 0, in, out, cyc); break;
                }
                hipDeviceSynchronize();
            }
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d wave(s)/SIMD  %-24s %.2f clock64 ticks per instruction per wave\n", waves, names[op], (double)c / (N * 16.0));
        }
    }
    return 0;
}
