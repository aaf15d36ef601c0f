// Microbenchmark: LDS atomic throughput on gfx950 (cycles per wave-instruction), random vs same-bank addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ void k(unsigned* out, long long* cyc, int iters, int tblmask) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i <= tblmask; i += blockDim.x) lds[i] = 0xFFFFFFFFu;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u, acc = 0;
    long long t0 = wall_clock64();
    long long c0 = clock64();
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        unsigned a = (x >> 8) & tblmask;
        if (OP == 0) acc += atomicCAS(&lds[a], 0xFFFFFFFFu, x);
        else if (OP == 1) acc += atomicMin(&lds[a], x);
        else if (OP == 2) atomicMin(&lds[a], x);           // no return
        else if (OP == 3) acc += lds[a];                   // plain read
        else if (OP == 4) lds[a] = x;                      // plain write
        else if (OP == 5) acc += atomicAdd(&lds[a], 1u);
    }
    long long c1 = clock64();
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = c1 - c0; cyc[blockIdx.x * 2 + 1] = t1 - t0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const int iters = 2000, tbl = 8192;
    unsigned* out; long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 1024 * 16);
    const char* names[] = {"CAS rtn", "min rtn", "min nortn", "read", "write", "add rtn"};
    for (int threads : {64, 256, 512, 1024}) {
        for (int op = 0; op < 6; op++) {
            auto launch = [&](int nb) {
                switch (op) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(nb), dim3(threads), tbl * 4, 0, out, cyc, iters, tbl - 1); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(nb), dim3(threads), tbl * 4, 0, out, cyc, iters, tbl - 1); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(nb), dim3(threads), tbl * 4, 0, out, cyc, iters, tbl - 1); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(nb), dim3(threads), tbl * 4, 0, out, cyc, iters, tbl - 1); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(nb), dim3(threads), tbl * 4, 0, out, cyc, iters, tbl - 1); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(nb), dim3(threads), tbl * 4, 0, out, cyc, iters, tbl - 1); break;
                }
            };
            launch(256); hipDeviceSynchronize();
            launch(256); hipDeviceSynchronize();
            std::vector<long long> h(512);
            hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost);
            double c = 0, t = 0; for (int i = 0; i < 256; i++) { c += h[2 * i]; t += h[2 * i + 1]; }
            c /= 256; t /= 256;
            int waves = threads / 64;
            printf("threads %4d  %-10s  %8.1f clk/iter/wave-set  -> %6.2f clk per wave-instr (CU level)  wall %.1f ns/iter\n", threads, names[op],
                   c / iters, c / iters / waves, t * 10.0 / iters);
        }
    }
    return 0;
}
