// Developer microbenchmark: issue cost of the VALU instructions the node step is made of, at 1, 2 and 4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o build/valu_issue tools/micro/valu_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
#define ITER 2000
template <int KIND>
__global__ void k(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = a, x2 = b, x3 = a + b, x4 = 1, x5 = 2, x6 = 3, x7 = 4;
    v2f p0{x0, x1}, p1{x2, x3}, p2{x4, x5}, p3{x6, x7};
    const v2f pa{a, a}, pb{b, b};
    long long t0 = clock64();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < REP / 4; r++) {
            if (KIND == 0) { // v_fma_f32, 4 independent chains
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
            } else if (KIND == 7) { // v_sub then v_mul plain (2+2)
                asm volatile("v_sub_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_sub_f32 %2, %2, %4\n v_mul_f32 %3, %3, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            }
        }
    }
    long long t1 = clock64();
    if (x0 + x1 + x2 + x3 + p0.x + p1.x + p2.x + p3.x + p0.y == 12345.678f) out[0] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}
template <int KIND> void run(const char* name, float* d) {
    for (int waves : {1, 2, 4}) { // waves per SIMD: one workgroup of 256 * waves threads on each CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<256, 256 * waves>>>(d, 1.0f, 0.5f);
        hipEventRecord(e0); k<KIND><<<256, 256 * waves>>>(d, 1.0f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        // s_memtime / clock64 ticks at 100 MHz constant; use wall time and the nominal 2.4 GHz instead
        const double cyc = ms * 1e-3 * 2.4e9, n = (double)ITER * REP;
        printf("%-28s waves/SIMD %d: %.2f cycles per instruction per wave, %.2f cycles per instruction per SIMD\n", name, waves, cyc / n, cyc / n / waves);
    }
}
int main() {
    float* d; hipMalloc(&d, 64);
    run<0>("v_fma_f32", d); run<1>("v_pk_mul_f32", d); run<2>("v_pk_add_f32", d); run<3>("v_max3_f32", d);
    run<4>("v_cmp+v_cndmask", d); run<5>("v_mul_f32 dependent", d); run<6>("s_add_u32", d); run<7>("v_sub/v_mul plain", d);
    return 0;
}
