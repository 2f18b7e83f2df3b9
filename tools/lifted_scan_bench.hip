// The candidate filter as ONE matrix product: the discriminant h^2 - a*c of a (sphere, ray) pair is bilinear in
// 11 per-sphere and 11 per-ray features (c_i c_j, c_i, 1, K  x  d_i d_j, ...), so with every feature split in two f16
// parts (hi*hi + hi*lo + lo*hi: 32 slots) a 32-sphere x 32-ray tile of discriminants is four chained
// v_mfma_f32_32x32x16_f16.  This bench times the inner loop such a scan would have -- per 32-sphere block: 4 x 16 B per
// lane of sphere features from LDS, 8 matrix instructions (two ray tiles), the maximum of the 32 results per lane, one
// compare and branch -- alone, beside a vector-only phase standing in for the rest of a path-tracing iteration, and that
// phase alone.  Workgroups of 512 threads share one 64 KB copy of the table; two of them per CU: four waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float F16v __attribute__((ext_vector_type(16)));
typedef _Float16 H8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BLOCKS = 16; // 16 blocks of 32 spheres: 512 spheres, 64 KB of f16 features

template <bool SCAN, bool OTHER>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void bench_kernel(const uint4 *__restrict__ table, float *out, unsigned long long *clocks, int iterations, int other_ops, float seed) {
    extern __shared__ uint4 l_table[]; // [block][chunk][lane]
    for (int i = threadIdx.x; i < BLOCKS * 4 * 64; i += 512) l_table[i] = table[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    H8 b[2][4];
    for (int t = 0; t < 2; ++t) for (int c = 0; c < 4; ++c) for (int j = 0; j < 8; ++j) b[t][c][j] = (_Float16)(seed * (float)((lane * 37 + t * 11 + c * 5 + j) % 64 - 32));
    float best = -1e30f, x0 = seed * lane, x1 = seed + lane, x2 = 1.0f + seed, x3 = 0.5f;
    int entered = 0;
    if constexpr (SCAN && OTHER) {
        // waves of one SIMD start at different points of the iteration
        const int wave = threadIdx.x >> 6, stagger = ((wave >> 2) * 2 + (blockIdx.x & 1)) * other_ops / 4 + (wave & 3) * other_ops / 16;
        for (int k = 0; k < stagger; k += 4) {
            x0 = __builtin_fmaf(x0, x2, x3); x1 = __builtin_fmaf(x1, x2, x0); x2 = __builtin_fmaf(x2, 0.999f, x3 * 1e-9f); x3 = __builtin_fmaxf(x3, x1 * 1e-30f);
        }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iterations; ++it) {
        if constexpr (SCAN) {
#pragma unroll 2
            for (int blk = 0; blk < BLOCKS; ++blk) {
                H8 a[4];
                for (int c = 0; c < 4; ++c) { const uint4 v = l_table[(blk * 4 + c) * 64 + lane]; __builtin_memcpy(&a[c], &v, 16); }
                F16v acc0 = {}, acc1 = {};
                for (int c = 0; c < 4; ++c) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[0][c], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[1][c], acc1, 0, 0, 0);
                }
                float g = -1e30f;
                for (int r = 0; r < 16; r += 1) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(g) : "v"(g), "v"(acc0[r]), "v"(acc1[r]));
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(g >= 1.0e30f) != 0ull, 0)) ++entered;
                best = __builtin_fmaxf(best, g);
            }
            b[0][0][0] += (_Float16)1.0f; // the next iteration is another ray
        }
        if constexpr (OTHER) {
            // dependent chains of plain vector instructions, four wide
            for (int k = 0; k < other_ops; k += 4) {
                x0 = __builtin_fmaf(x0, x2, x3); x1 = __builtin_fmaf(x1, x2, x0); x2 = __builtin_fmaf(x2, 0.999f, x3 * 1e-9f); x3 = __builtin_fmaxf(x3, x1 * 1e-30f);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 7) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
    if (best + x0 + x1 + x2 + x3 == 123.456f || entered == 77) out[0] = best;
}

static unsigned long long *g_clocks;
template <class K> double run(const char *name, K kernel, const uint4 *table, float *out, int other_ops) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BLOCKS * 4 * 64 * 16));
    const int iterations = 20000, grid = 256 * 2;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), BLOCKS * 4 * 64 * 16, 0, table, out, g_clocks, 100, other_ops, -0.001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), BLOCKS * 4 * 64 * 16, 0, table, out, g_clocks, iterations, other_ops, -0.001f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns_per_iteration_per_simd = ms * 1e6 / ((double)iterations * 4); // four waves per SIMD, one after the other
    unsigned long long c[2]; CHECK(hipMemcpy(c, g_clocks, 16, hipMemcpyDeviceToHost));
    printf("%-44s %8.0f ns per wave iteration per SIMD at %.2f GHz", name, ns_per_iteration_per_simd, (double)c[0] / ((double)c[1] * 10.0));
    return ns_per_iteration_per_simd;
}

int main() {
    uint4 *table; float *out;
    const size_t bytes = (size_t)BLOCKS * 4 * 64 * 16;
    CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&g_clocks, 16));
    unsigned int *h = (unsigned int *)malloc(bytes);
    unsigned int x = 12345u;
    for (size_t i = 0; i < bytes / 4; ++i) { // random f16 pairs of ordinary size, both signs
        x = x * 1664525u + 1013904223u; const unsigned int hi = (0x3000u + ((x >> 16) & 0xfffu)) ^ ((x >> 3) & 0x8000u);
        x = x * 1664525u + 1013904223u; const unsigned int lo = (0x3000u + ((x >> 16) & 0xfffu)) ^ ((x >> 3) & 0x8000u);
        h[i] = (hi << 16) | lo;
    }
    CHECK(hipMemcpy(table, h, bytes, hipMemcpyHostToDevice));
    const double scan = run("matrix scan of 512 spheres alone", bench_kernel<true, false>, table, out, 0);
    printf("  = %.1f ns per 32-sphere block (vector scan: 270)\n", scan / BLOCKS);
    for (int other : {1000, 2000, 4000}) {
        char name[96];
        snprintf(name, sizeof name, "%d vector instructions alone", other);
        const double o = run(name, bench_kernel<false, true>, table, out, other); printf("\n");
        snprintf(name, sizeof name, "matrix scan + %d vector instructions", other);
        const double both = run(name, bench_kernel<true, true>, table, out, other);
        printf("  (sum of the two alone: %.0f)\n", scan + o);
        (void)both;
    }
    return 0;
}
