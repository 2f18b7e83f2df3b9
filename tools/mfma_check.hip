// Checks on the GPU that v_mfma_f32_16x16x4_f32 is bit-for-bit the k-ordered chain of fused
// multiply-adds  D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))  that the scan's VALU
// path and the CPU oracle evaluate, on data shaped like the scan's (sphere records against
// ray vectors), and times the instruction.  Prints mismatches (expected 0).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// A: 16 rows x 4 (row-major), B: 4 x 16 (row-major), C: 16 x 16.  One wave.
__global__ void mfma_once(const float *A, const float *B, const float *C, float *D) {
    const int l = threadIdx.x;
    const float a = A[(l & 15) * 4 + (l >> 4)];  // A[i = l&15][k = l>>4]
    const float b = B[(l >> 4) * 16 + (l & 15)]; // B[k = l>>4][j = l&15]
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * (l >> 4) + r) * 16 + (l & 15)];
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = d[r];
}

__global__ void mfma_rate(float *out, float a, float b, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + i, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

int main() {
    float *dA, *dB, *dC, *dD;
    CHECK(hipMalloc(&dA, 64 * 4)); CHECK(hipMalloc(&dB, 64 * 4)); CHECK(hipMalloc(&dC, 256 * 4)); CHECK(hipMalloc(&dD, 256 * 4));
    std::vector<float> A(64), B(64), C(256), D(256);
    uint64_t state = 88172645463325252ull;
    auto rnd = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (double)(state >> 11) / 9007199254740992.0; };
    long mismatches = 0, checked = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        const int kind = trial % 4;
        for (int i = 0; i < 16; ++i) { // rows: sphere records {cx, cy, cz, K}
            const double scale = kind == 0 ? 12.0 : kind == 1 ? 1000.0 : kind == 2 ? 0.5 : 1e-3;
            A[i * 4 + 0] = (float)((rnd() * 2 - 1) * scale); A[i * 4 + 1] = (float)((rnd() * 2 - 1) * scale);
            A[i * 4 + 2] = (float)((rnd() * 2 - 1) * scale); A[i * 4 + 3] = (float)((rnd() * 2 - 1) * scale * scale);
        }
        for (int j = 0; j < 16; ++j) { // columns: ray vectors {x, y, z, 1 or 0}
            B[0 * 16 + j] = (float)((rnd() * 2 - 1) * 26.0); B[1 * 16 + j] = (float)((rnd() * 2 - 1) * 4.0);
            B[2 * 16 + j] = (float)((rnd() * 2 - 1) * 26.0); B[3 * 16 + j] = (trial & 1) ? 1.0f : 0.0f;
            const float cin = (float)(rnd() * 400.0 - 100.0);
            for (int i = 0; i < 16; ++i) C[i * 16 + j] = cin;
        }
        CHECK(hipMemcpy(dA, A.data(), 64 * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), 64 * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dC, C.data(), 256 * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        CHECK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float ref = C[i * 16 + j];
                for (int k = 0; k < 4; ++k) ref = std::fmaf(A[i * 4 + k], B[k * 16 + j], ref);
                ++checked;
                if (std::memcmp(&ref, &D[i * 16 + j], 4) != 0) {
                    if (mismatches < 5) printf("mismatch trial %d (%d,%d): mfma %.9g chain %.9g\n", trial, i, j, D[i * 16 + j], ref);
                    ++mismatches;
                }
            }
    }
    printf("mfma_f32_16x16x4f32 vs k-ordered fmaf chain: %ld mismatches in %ld values\n", mismatches, checked);

    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4}) {
        const int iters = 20000, grid = 256 * wps;
        hipLaunchKernelGGL(mfma_rate, dim3(grid), dim3(256), 0, 0, dD, 1.0f, 1.0001f, 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_rate, dim3(grid), dim3(256), 0, 0, dD, 1.0f, 1.0001f, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)grid * 4 * iters * 8 * 2048.0;
        printf("waves/SIMD %d: %.1f TFLOP/s f32 MFMA (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wps, flops / (ms * 1e-3) * 1e-12, (ms * 1e-3) * 2.4e9 / ((double)iters * 8 * wps));
    }
    return mismatches != 0;
}
