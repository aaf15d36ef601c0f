#include <hip/hip_runtime.h>
// Reduced form of the loop that hung the GPU in round 3 (route.hip.h, long lists, phase C): a wave takes groups off an LDS counter
// with `if (lane == 0) g = atomicAdd(..); g = __shfl(g, 0)` inside a loop that leaves through break / continue.  Not part of the
// library; kept so the code the compiler makes of the idiom can be looked at again:
//     hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/lane0_loop.s tools/micro/lane0_loop.hip
// What the ISA shows (ROCm 7.2, gfx950), and what DESIGN.md §3.2a records:
//   * the loop is structurised with ONE accumulated "has left" mask: at the latch `s_and_b64 s[4:5], exec, <break cond>;
//     s_or_b64 s[10:11], s[4:5], s[10:11]; s_andn2_b64 exec, exec, s[10:11]; s_cbranch_execz <exit>` — lanes leave the loop ONE BY
//     ONE as far as the code is concerned, the wave stays in it until EXEC is empty;
//   * the header runs `s_and_saveexec_b64 .., vcc` (vcc = lane == 0, hoisted) around the atomic — no lane there, no atomic —
//     and then `ds_bpermute_b32 v4, v21, v4` with the address of lane 0: a bpermute reads 0 from a source lane that is not in
//     EXEC.  So the moment lane 0 has left the loop and any other lane has not, every remaining lane reads g = 0, takes group 0
//     again, and nothing ever advances the counter: the wave never finishes (the GPU hang).
//   * hence the idiom is safe exactly when every exit of the loop is WAVE-UNIFORM AT RUN TIME (all lanes leave in the same trip);
//     a per-lane exit before the broadcast — in round 3 most likely the limit test taken on a lane's own element rank instead
//     of the group's start — turns it into the hang.  The library now hands groups out statically per wave (no counter), and
//     the three remaining `lane == 0` atomic + broadcast sites (route.hip.h compaction, route_lazy.hip.h LZ_INSERT and the
//     crossing-level insert) sit in loops whose trip counts are computed from LDS / scalar values behind a barrier: uniform.
__global__ void k(const int* __restrict__ bins, const int* __restrict__ cursor, int ngrp, int nout, int wcap, int* out) {
    __shared__ int s_next;
    __shared__ unsigned slice[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();
    int acc = 0;
    for (;;) {
        int g;
        if (lane == 0) g = atomicAdd(&s_next, 1);
        g = __shfl(g, 0);
        if (g >= ngrp) break;
        const int c = bins[g];
        const int g0 = cursor[g] - c;
        if (c > 0 && g0 >= nout) break;
        if (c == 0 || c > wcap) continue;
        for (int i = lane; i < c; i += 64) slice[wave][i] = g0 + i;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < c; i += 64) { int rk = 0; for (int j = 0; j < c; j++) rk += slice[wave][j] < slice[wave][i]; acc += rk; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
