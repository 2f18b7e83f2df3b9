// Isolates one step of the matrix-core block filter: 8 x v_mfma_f32_16x16x4_f32 (16 spheres x 64
// rays) + the 32 discriminants and their maximum, and times it per SIMD at several occupancies.
// Variants: accumulators in AGPRs (what hipcc picks) or in VGPRs (inline asm).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float F4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(256) void step_kernel(const float *__restrict__ rows, float *out, int blocks, float seed) {
    const int lane = threadIdx.x & 63;
    float bh[4], bc[4], ag[4];
    F4 ch[4], cc[4];
    for (int g = 0; g < 4; ++g) {
        bh[g] = seed * (lane + g); bc[g] = seed * (lane - g); ag[g] = 1.0f + g;
        ch[g] = F4{seed, seed, seed, seed}; cc[g] = F4{seed + g, seed + g, seed + g, seed + g};
    }
    float best = -1e30f;
    float row = rows[lane];
    for (int b = 0; b < blocks; ++b) {
        const float next = rows[((b + 1) & 31) * 64 + lane];
        F4 hh[4], cv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (VARIANT == 0) {
                hh[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(row, bh[g], ch[g], 0, 0, 0);
                cv[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(row, bc[g], cc[g], 0, 0, 0);
            } else {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=v"(hh[g]) : "v"(row), "v"(bh[g]), "v"(ch[g]));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=v"(cv[g]) : "v"(row), "v"(bc[g]), "v"(cc[g]));
            }
        }
        if (VARIANT == 1) asm volatile("s_nop 7\n s_nop 7" ::: "memory"); // results of the last MFMA must have landed
        float gate[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float d0 = __builtin_fmaf(hh[g][0], hh[g][0], -(ag[g] * cv[g][0])), d1 = __builtin_fmaf(hh[g][1], hh[g][1], -(ag[g] * cv[g][1]));
            const float d2 = __builtin_fmaf(hh[g][2], hh[g][2], -(ag[g] * cv[g][2])), d3 = __builtin_fmaf(hh[g][3], hh[g][3], -(ag[g] * cv[g][3]));
            gate[g] = __builtin_fmaxf(__builtin_fmaxf(d0, d1), __builtin_fmaxf(d2, d3));
        }
        best = __builtin_fmaxf(best, __builtin_fmaxf(__builtin_fmaxf(gate[0], gate[1]), __builtin_fmaxf(gate[2], gate[3])));
        row = next;
    }
    if (best == 123.456f) out[0] = best;
}

template <class K> void run(const char *name, K kernel, const float *rows, float *out) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 20000, grid = 256 * wps;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, rows, out, 10, 0.001f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, rows, out, blocks, 0.001f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double cycles_per_block_per_simd = ms * 1e-3 * 2.4e9 / ((double)blocks * wps);
        printf("%s waves/SIMD %d: %.0f cycles per 16-sphere block per SIMD (matrix pipe alone: 256), %.2f tests/cycle/SIMD\n", name, wps, cycles_per_block_per_simd, 1024.0 / cycles_per_block_per_simd);
    }
}

int main() {
    float *rows, *out;
    CHECK(hipMalloc(&rows, 33 * 64 * 4)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(rows, 0, 33 * 64 * 4));
    run("AGPR accumulators", step_kernel<0>, rows, out);
    run("VGPR accumulators", step_kernel<1>, rows, out);
    return 0;
}
