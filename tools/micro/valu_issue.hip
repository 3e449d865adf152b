// Developer microbenchmark: issue cost of the VALU instructions the node step is made of, at 1, 2, 3 and 4 waves per SIMD,
// in TRUE shader cycles.
// build: hipcc -O3 --offload-arch=gfx950 -o build/valu_issue tools/micro/valu_issue.hip ; run on the GPU box.
//
// Round 2 turned wall time into cycles with the nominal 2.4 GHz and read v_fma_f32 at 4.0 cycles per SIMD where the guide
// says 2 (MI355X_MICROARCH.md, constants table): the clock was an assumption.  This version stamps s_memtime (one tick =
// one shader cycle, guide :488) and s_memrealtime (100 MHz) around the loop in every workgroup and reports
//   cycles per instruction per SIMD = median over workgroups of (LONGEST delta s_memtime among the workgroup's waves) / instructions / waves per SIMD
//   in-kernel clock                 = delta s_memtime / delta s_memrealtime x 100 MHz
// next to the wall-clock figure, so that clock and issue cost are separated.  Every wave stamps: the oldest wave of a SIMD wins the
// issue arbitration and runs its loop at its single-wave pace whatever shares the SIMD (its own stamp reads 5.3 cycles per
// instruction at 1, 2, 3 and 4 waves per SIMD); the SIMD's cost per instruction is what the LAST wave to finish sees.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
#define ITER 4000
template <int KIND>
__global__ void k(float* out, unsigned long long* stamps, float a, float b) {
    float x0 = threadIdx.x, x1 = a, x2 = b, x3 = a + b, x4 = 1, x5 = 2, x6 = 3, x7 = 4;
    v2f p0{x0, x1}, p1{x2, x3}, p2{x4, x5}, p3{x6, x7};
    const v2f pa{a, a}, pb{b, b};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < REP / 4; r++) {
            if (KIND == 0) { // v_fma_f32, 4 independent chains, three VGPR sources
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            } else if (KIND == 1) { // v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            } else if (KIND == 2) { // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
            } else if (KIND == 3) { // v_max3_f32
                asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            } else if (KIND == 4) { // v_cmp + v_cndmask pairs
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "vcc");
            } else if (KIND == 5) { // v_mul_f32 dependent chain (latency)
                asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(a));
            } else if (KIND == 6) { // s_ ops (SALU) 4
                int s; asm volatile("s_add_u32 %0, 1, 2\n s_add_u32 %0, %0, 2\n s_add_u32 %0, %0, 2\n s_add_u32 %0, %0, 2" : "=s"(s) : : "scc");
            } else if (KIND == 7) { // v_sub then v_mul plain (VOP2)
                asm volatile("v_sub_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_sub_f32 %2, %2, %4\n v_mul_f32 %3, %3, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            } else if (KIND == 8) { // v_fma_f32 with two SGPR... one SGPR source (constant bus), one VGPR: the scalar-row form of the node step
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(a), "v"(b));
            } else if (KIND == 9) { // v_mul_f32 independent (VOP2), 4 chains
                asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
            } else if (KIND == 10) { // v_fma_f32 with two distinct VGPR sources only (x = x * a + a)
                asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
            } else if (KIND == 11) { // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            } else if (KIND == 12) { // v_min_f32 / v_max_f32 (VOP2)
                asm volatile("v_min_f32 %0, %0, %4\n v_max_f32 %1, %1, %5\n v_min_f32 %2, %2, %4\n v_max_f32 %3, %3, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (x0 + x1 + x2 + x3 + p0.x + p1.x + p2.x + p3.x + p0.y == 12345.678f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * 16 + threadIdx.x / 64; stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0; }
}
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
template <int KIND> void run(const char* name, float* d, unsigned long long* stamps) {
    const int n_wg = 256;
    for (int waves : {1, 2, 3, 4}) { // waves per SIMD: one workgroup of 256 * waves threads on each CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; w++) k<KIND><<<n_wg, 256 * waves>>>(d, stamps, 1.0f, 0.5f); // warm the clock
        hipEventRecord(e0); k<KIND><<<n_wg, 256 * waves>>>(d, stamps, 1.0f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * n_wg * 16);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int b = 0; b < n_wg; b++) {
            int slow = b * 16;
            for (int w = 0; w < 4 * waves; w++) if (h[2 * (b * 16 + w)] > h[2 * slow]) slow = b * 16 + w;
            cyc.push_back((double)h[2 * slow]); ghz.push_back((double)h[2 * slow] / (double)h[2 * slow + 1] * 0.1);
        }
        const double n = (double)ITER * REP, c = median(cyc), f = median(ghz);
        printf("%-34s waves/SIMD %d: %5.2f cycles per instruction per SIMD (s_memtime), in-kernel clock %.3f GHz, wall %.3f ms = %.2f cycles per SIMD at that clock\n",
               name, waves, c / n / waves, f, ms, ms * 1e-3 * f * 1e9 / n / waves);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}
int main() {
    float* d; hipMalloc(&d, 64);
    unsigned long long* stamps; hipMalloc(&stamps, 2 * 256 * 16 * 8);
    run<0>("v_fma_f32 (3 VGPR sources)", d, stamps); run<10>("v_fma_f32 (2 distinct VGPR src)", d, stamps); run<8>("v_fma_f32 (SGPR + 2 VGPR src)", d, stamps);
    run<9>("v_mul_f32 independent (VOP2)", d, stamps); run<7>("v_sub/v_mul plain (VOP2)", d, stamps); run<12>("v_min/v_max (VOP2)", d, stamps);
    run<1>("v_pk_mul_f32", d, stamps); run<2>("v_pk_add_f32", d, stamps); run<11>("v_pk_fma_f32", d, stamps); run<3>("v_max3_f32", d, stamps);
    run<4>("v_cmp+v_cndmask", d, stamps); run<5>("v_mul_f32 dependent", d, stamps); run<6>("s_add_u32", d, stamps);
    return 0;
}
