// Can the candidate filter use BOTH pipes of a SIMD at once?  One wave issues, per block, the eight
// v_mfma_f32_16x16x4_f32 of 16 spheres x 64 rays (h and a*c, bit-identical to the fused chains of the vector scan)
// interleaved one by one with (i) the vector work on the PREVIOUS block's matrix results (16 discriminants, their
// maximum, one compare) and (ii) the packed vector scan of EXTRA more sphere pairs.  Prints the time per sphere per
// SIMD against the vector scan alone (20 shader cycles per sphere in the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef float F4 __attribute__((ext_vector_type(4)));
typedef float F2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ F2 fma2(F2 a, F2 b, F2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int EXTRA, bool MATRIX, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void hybrid_kernel(const float *__restrict__ table, float *out, unsigned long long *clocks, int blocks, float seed) {
    const int lane = threadIdx.x & 63;
    float bh[4], bm[4];
    F4 ch[4], cm[4];
    for (int t = 0; t < 4; ++t) {
        bh[t] = seed * (lane * 37 % 64 + t); bm[t] = seed * (lane * 29 % 64 - t);
        ch[t] = F4{seed, seed, seed, seed} * (float)(t + 1); cm[t] = F4{seed, seed, seed, seed} * (float)(t + 5);
    }
    F4 oh[2][4], om[2][4];
    for (int t = 0; t < 4; ++t) { oh[0][t] = oh[1][t] = ch[t]; om[0][t] = om[1][t] = cm[t]; }
    // the vector scan's per-ray constants
    const F2 dx2 = {seed * lane, seed * lane}, dy2 = {seed + lane, seed + lane}, dz2 = {1.0f + seed, 1.0f + seed}, nd_o2 = {0.5f, 0.5f};
    const F2 ox2 = {seed * 3.0f, seed * 3.0f}, oy2 = {seed * 5.0f, seed * 5.0f}, oz2 = {seed * 7.0f, seed * 7.0f}, oo2 = {2.0f, 2.0f}, a2 = {1.5f, 1.5f};
    typedef const float __attribute__((address_space(4))) *ConstF;
    ConstF c_tab = (ConstF)table;
    float best = -1e30f;
    int entered = 0;
    float arow = table[lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    auto vector_pair = [&](const int j) { // two spheres on the vector pipe, records through the scalar cache
        const F2 cx = {c_tab[8 * j], c_tab[8 * j + 1]}, cy = {c_tab[8 * j + 2], c_tab[8 * j + 3]}, cz = {c_tab[8 * j + 4], c_tab[8 * j + 5]}, k = {c_tab[8 * j + 6], c_tab[8 * j + 7]};
        const F2 h = fma2(cz, dz2, fma2(cy, dy2, fma2(cx, dx2, nd_o2)));
        const F2 m = fma2(k, a2, fma2(cz, oz2, fma2(cy, oy2, fma2(cx, ox2, oo2))));
        const F2 d = fma2(h, h, -m);
        return __builtin_fmaxf(d.x, d.y);
    };
    auto block = [&](const int b, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value, prev = cur ^ 1;
        const float anext = table[((b + 1) & 31) * 64 + lane];
        float g = -1e30f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if constexpr (MATRIX) {
                oh[cur][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow, bh[t], ch[t], 0, 0, 0);
                om[cur][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow, bm[t], cm[t], 0, 0, 0);
                // the previous block's tile t: four discriminants and their maximum
                const F4 h = oh[prev][t], m = om[prev][t];
                const float d0 = __builtin_fmaf(h[0], h[0], -m[0]), d1 = __builtin_fmaf(h[1], h[1], -m[1]);
                const float d2 = __builtin_fmaf(h[2], h[2], -m[2]), d3 = __builtin_fmaf(h[3], h[3], -m[3]);
                float g1, g2;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(g1) : "v"(g), "v"(d0), "v"(d1));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(g2) : "v"(g1), "v"(d2), "v"(d3));
                g = g2;
            }
#pragma unroll
            for (int e = t; e < EXTRA; e += 4) g = __builtin_fmaxf(g, vector_pair(((b * EXTRA + e) & 63)));
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(g >= 1.0e30f) != 0ull, 0)) ++entered;
        best = __builtin_fmaxf(best, g);
        arow = anext;
    };
    for (int b = 0; b < blocks; b += 2) {
        block(b, std::integral_constant<int, 0>{});
        block(b + 1, std::integral_constant<int, 1>{});
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 5) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
    if (best == 123.456f || entered == 77) out[0] = best + oh[0][0][0] + om[1][3][3];
}

static unsigned long long *g_clocks;
template <class K> void run(const char *name, K kernel, int waves, int spheres_per_block, const float *table, float *out) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 100000, grid = 256 * waves;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, table, out, g_clocks, 1000, -0.001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, table, out, g_clocks, blocks, -0.001f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c[2]; CHECK(hipMemcpy(c, g_clocks, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)c[0] / ((double)c[1] * 10.0);
    const double ns_per_block = ms * 1e6 / ((double)blocks * waves);
    printf("%-40s %d waves/SIMD: %6.1f ns per block of %2d spheres = %5.2f ns = %5.1f cycles per sphere per SIMD at %.2f GHz (vector scan in the kernel: 8.4 ns, 20 cycles)\n",
           name, waves, ns_per_block, spheres_per_block, ns_per_block / spheres_per_block, ns_per_block / spheres_per_block * ghz, ghz);
}

int main() {
    float *table, *out;
    CHECK(hipMalloc(&table, 33 * 64 * 4)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&g_clocks, 16));
    float *h = (float *)malloc(33 * 64 * 4);
    unsigned int x = 12345u;
    for (int i = 0; i < 33 * 64; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 21)); }
    CHECK(hipMemcpy(table, h, 33 * 64 * 4, hipMemcpyHostToDevice));
#define RUN(EXTRA, MATRIX, WAVES, label) run(label, hybrid_kernel<EXTRA, MATRIX, WAVES>, WAVES, (MATRIX ? 16 : 0) + 2 * EXTRA, table, out)
    RUN(8, false, 4, "vector scan alone, 8 pairs per block");
    RUN(0, true, 2, "matrix, 16 spheres");
    RUN(0, true, 3, "matrix, 16 spheres");
    RUN(2, true, 2, "matrix 16 + vector 4 spheres");
    RUN(2, true, 3, "matrix 16 + vector 4 spheres");
    RUN(4, true, 2, "matrix 16 + vector 8 spheres");
    RUN(4, true, 3, "matrix 16 + vector 8 spheres");
    RUN(6, true, 2, "matrix 16 + vector 12 spheres");
    RUN(6, true, 3, "matrix 16 + vector 12 spheres");
    RUN(8, true, 2, "matrix 16 + vector 16 spheres");
    RUN(8, true, 3, "matrix 16 + vector 16 spheres");
    return 0;
}
