// Measures the FP32 vector rate of the chip with dependent-free v_fma_f32 and v_pk_fma_f32
// streams at several occupancies, and the in-kernel clock, so that the roofline's peak can
// be quoted next to a number measured on the same box.
//   usage: valu_peak            (prints one line per variant, then a JSON summary line)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ACC = 16;   // independent accumulators per lane
constexpr int ITERS = 4096;

__global__ void fma_scalar(float *out, float b, float c) {
    float acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = threadIdx.x * 1e-6f + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(b), "v"(c));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    if (s == 123.456f) out[0] = s; // never true: keeps the chain alive
}

// the same stream with one operand of every fma in an SGPR (how the scan reads sphere records)
__global__ void fma_sgpr(float *out, float b, float c) {
    float acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = threadIdx.x * 1e-6f + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "s"(b), "v"(c));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    if (s == 123.456f) out[0] = s;
}

// VOP2 encoding with an SGPR source: v_fmac_f32 acc, s, v
__global__ void fmac_sgpr(float *out, float b, float c) {
    float acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = threadIdx.x * 1e-6f + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "s"(b), "v"(c));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    if (s == 123.456f) out[0] = s;
}

// packed fma with one operand an SGPR pair (two spheres' worth of a record component)
__global__ void fma_packed_sgpr(float *out, float b, float c) {
    v2f acc[ACC / 2];
    const v2f vb = {b, b * 1.0001f}, vc = {c, c * 0.9999f};
#pragma unroll
    for (int i = 0; i < ACC / 2; ++i) acc[i] = v2f{threadIdx.x * 1e-6f + i, threadIdx.x * 2e-6f + i};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC / 2; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "s"(vb), "v"(vc));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC / 2; ++i) s += acc[i].x + acc[i].y;
    if (s == 123.456f) out[0] = s;
}

__global__ void fma_packed(float *out, float b, float c) {
    v2f acc[ACC / 2];
    const v2f vb = {b, b * 1.0001f}, vc = {c, c * 0.9999f};
#pragma unroll
    for (int i = 0; i < ACC / 2; ++i) acc[i] = v2f{threadIdx.x * 1e-6f + i, threadIdx.x * 2e-6f + i};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(vb), "v"(vc));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC / 2; ++i) s += acc[i].x + acc[i].y;
    if (s == 123.456f) out[0] = s;
}

// 3 sub + 6 fma + 2 mul-class: the instruction mix of the ray/sphere test, no memory, no branch
__global__ void sphere_mix(float *out, float b, float c) {
    float ox = threadIdx.x * 1e-3f, oy = 0.5f, oz = -0.25f, dx = 0.3f, dy = b, dz = c, a = dx * dx + dy * dy + dz * dz;
    float best = 0.f;
    for (int it = 0; it < ITERS; ++it) {
        float cx = it * 1e-3f, cy = b * it, cz = c - it;
        float x = cx - ox, y = cy - oy, z = cz - oz;
        float h = __builtin_fmaf(dz, z, __builtin_fmaf(dy, y, dx * x));
        float cc = __builtin_fmaf(z, z, __builtin_fmaf(y, y, __builtin_fmaf(x, x, -0.04f)));
        float disc = __builtin_fmaf(h, h, -(a * cc));
        best = disc > best ? disc : best;
    }
    if (best == 123.456f) out[0] = best;
}

// issue cost of the instructions the non-scan part of the renderer is made of (RNG hash, square
// roots, reciprocals): the same dependent-free stream with one instruction swapped in
#define OP_KERNEL(name, text, constraint_b)                                                         \
    __global__ void name(float *out, float b, float c) {                                            \
        uint32_t acc[ACC];                                                                          \
        const uint32_t ub = __float_as_uint(b) | 1u;                                                \
        (void)c;                                                                                    \
        _Pragma("unroll") for (int i = 0; i < ACC; ++i) acc[i] = __float_as_uint(1.0f + threadIdx.x * 1e-3f + i); \
        for (int it = 0; it < ITERS; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < ACC; ++i) asm volatile(text : "+v"(acc[i]) : constraint_b(ub)); \
        }                                                                                           \
        uint32_t s = 0;                                                                             \
        _Pragma("unroll") for (int i = 0; i < ACC; ++i) s += acc[i];                                \
        if (s == 0x12345678u) out[0] = 1.0f;                                                        \
    }
OP_KERNEL(op_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1", "v")
OP_KERNEL(op_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1", "v")
OP_KERNEL(op_xor, "v_xor_b32 %0, %0, %1", "v")
OP_KERNEL(op_sqrt, "v_sqrt_f32 %0, %0", "v")
OP_KERNEL(op_rcp, "v_rcp_f32 %0, %0", "v")
OP_KERNEL(op_cvt, "v_cvt_f32_u32 %0, %0", "v")
OP_KERNEL(op_max, "v_max_f32 %0, %0, %1", "v")


// packed fma whose scalar-pair operand is selected per half (op_sel): the low dword for both halves; the same with the
// scalar pair as multiplier AND addend (one scalar operand read twice); and the plain fma that closes such a chain,
// reading both its factors from the low half of a pair and adding into the high half
#define PK_KERNEL(name, text)                                                                       \
    __global__ void name(float *out, float b, float c) {                                            \
        v2f acc[ACC / 2];                                                                           \
        const v2f vb = {b, b * 1.0001f}, vc = {c, c * 0.9999f};                                     \
        _Pragma("unroll") for (int i = 0; i < ACC / 2; ++i) acc[i] = v2f{threadIdx.x * 1e-6f + i, threadIdx.x * 2e-6f + i}; \
        for (int it = 0; it < ITERS; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < ACC / 2; ++i) asm volatile(text : "+v"(acc[i]) : "s"(vb), "v"(vc)); \
        }                                                                                           \
        float s = 0.f;                                                                              \
        _Pragma("unroll") for (int i = 0; i < ACC / 2; ++i) s += acc[i].x + acc[i].y;               \
        if (s == 123.456f) out[0] = s;                                                              \
    }
PK_KERNEL(pk_sel_lo, "v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]")
PK_KERNEL(pk_sel_hi, "v_pk_fma_f32 %0, %1, %0, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]")
PK_KERNEL(pk_same_pair, "v_pk_fma_f32 %0, %1, %0, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]")
PK_KERNEL(pk_vgpr_sel, "v_pk_fma_f32 %0, %2, %0, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]")
__global__ void fmac_halves(float *out, float b, float c) {
    v2f acc[ACC];
    (void)b; (void)c;
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = v2f{threadIdx.x * 1e-6f + i, threadIdx.x * 2e-6f + i};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(acc[i].y) : "v"(acc[i].x));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i].x + acc[i].y;
    if (s == 123.456f) out[0] = s;
}

// the instruction mix of a one-sphere chain: three packed operations and one plain one, on independent registers
__global__ void mix_3pk_1fma(float *out, float b, float c) {
    v2f acc[6];
    float pl[2];
    const v2f vb = {b, b * 1.0001f}, vc = {c, c * 0.9999f};
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = v2f{threadIdx.x * 1e-6f + i, threadIdx.x * 2e-6f + i};
    pl[0] = b; pl[1] = c;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[3 * i]) : "s"(vb), "v"(vc));
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[3 * i + 1]) : "s"(vb), "v"(vc));
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[3 * i + 2]) : "s"(vb), "v"(vc));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(pl[i]) : "v"(b), "v"(c));
        }
    }
    float s = pl[0] + pl[1];
#pragma unroll
    for (int i = 0; i < 6; ++i) s += acc[i].x + acc[i].y;
    if (s == 123.456f) out[0] = s;
}

template <class K> double run(K kernel, int blocks_per_cu, int threads, float *d) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, 0, d, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, 0, d, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) best = ms / 10;
    }
    return best * 1e-3; // seconds per launch
}

int main() {
    float *d; CHECK(hipMalloc(&d, 4096));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    double best_scalar = 0, best_packed = 0, best_mix = 0;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        // 256-thread blocks = 1 wave per SIMD per block; blocks_per_cu = waves per SIMD
        const double flops = 256.0 * waves_per_simd * 256 * double(ITERS) * ACC * 2;
        double ts = run(fma_scalar, waves_per_simd, 256, d);
        double tp = run(fma_packed, waves_per_simd, 256, d);
        double tsg = run(fma_sgpr, waves_per_simd, 256, d);
        double tpsg = run(fma_packed_sgpr, waves_per_simd, 256, d);
        double tsc = run(fmac_sgpr, waves_per_simd, 256, d);
        const double mix_flops = 256.0 * waves_per_simd * 256 * double(ITERS) * 17;
        double tm = run(sphere_mix, waves_per_simd, 256, d);
        printf("waves/SIMD %d: v_fma_f32 %.1f TFLOP/s   v_pk_fma_f32 %.1f   v_fma_f32 (SGPR src) %.1f   v_fmac_f32 (SGPR src) %.1f   v_pk_fma_f32 (SGPR pair src) %.1f   sphere-mix %.1f (17 FLOP/test)\n", waves_per_simd,
               flops / ts * 1e-12, flops / tp * 1e-12, flops / tsg * 1e-12, flops / tsc * 1e-12, flops / tpsg * 1e-12, mix_flops / tm * 1e-12);
        if (flops / ts > best_scalar) best_scalar = flops / ts;
        if (flops / tp > best_packed) best_packed = flops / tp;
        if (mix_flops / tm > best_mix) best_mix = mix_flops / tm;
    }
    {
        // cycles per wave64 instruction on one SIMD at 8 waves per SIMD, clock from the device properties
        const double insts = 256.0 * 8 * 4 * double(ITERS) * ACC; // wave-instructions per launch: CUs * blocks * waves * ...
        const double hz = prop.clockRate * 1e3;
        struct { const char *name; double t; } ops[] = {
            {"v_xor_b32", run(op_xor, 8, 256, d)}, {"v_max_f32", run(op_max, 8, 256, d)}, {"v_cvt_f32_u32", run(op_cvt, 8, 256, d)},
            {"v_mul_u32_u24", run(op_mul_u32_u24, 8, 256, d)}, {"v_mul_lo_u32", run(op_mul_lo_u32, 8, 256, d)},
            {"v_sqrt_f32", run(op_sqrt, 8, 256, d)}, {"v_rcp_f32", run(op_rcp, 8, 256, d)}};
        printf("cycles per wave64 instruction per SIMD (8 waves/SIMD):");
        for (auto &o : ops) printf("  %s %.1f", o.name, o.t * hz * 1024.0 / insts);
        printf("\n");
    }
    for (int waves_per_simd : {4, 8}) {
        // cycles per wave64 instruction per SIMD of the operand-selected forms
        const double hz = prop.clockRate * 1e3;
        const double insts_pk = 256.0 * waves_per_simd * 4 * double(ITERS) * (ACC / 2), insts = insts_pk * 2;
        printf("waves/SIMD %d, cycles per wave64 instruction per SIMD: v_pk_fma_f32 (SGPR pair) %.2f  op_sel low half %.2f  op_sel high half %.2f  same SGPR pair as factor and addend %.2f  VGPR pair as factor and addend, op_sel %.2f  v_fmac_f32 on the halves of a pair %.2f\n",
               waves_per_simd, run(fma_packed_sgpr, waves_per_simd, 256, d) * hz / (insts_pk / 1024.0), run(pk_sel_lo, waves_per_simd, 256, d) * hz / (insts_pk / 1024.0),
               run(pk_sel_hi, waves_per_simd, 256, d) * hz / (insts_pk / 1024.0), run(pk_same_pair, waves_per_simd, 256, d) * hz / (insts_pk / 1024.0),
               run(pk_vgpr_sel, waves_per_simd, 256, d) * hz / (insts_pk / 1024.0), run(fmac_halves, waves_per_simd, 256, d) * hz / (insts / 1024.0));
    }
    for (int waves_per_simd : {4, 8}) {
        const double hz = prop.clockRate * 1e3;
        const double groups = 256.0 * waves_per_simd * 4 * double(ITERS) * 2; // groups of 3 packed + 1 plain per launch
        printf("waves/SIMD %d: 3 v_pk_fma_f32 + 1 v_fma_f32 cost %.2f cycles per SIMD (sum of the single-instruction costs: 15.3)\n", waves_per_simd, run(mix_3pk_1fma, waves_per_simd, 256, d) * hz / (groups / 1024.0));
    }
    printf("{\"v_fma_f32_tflops\": %.2f, \"v_pk_fma_f32_tflops\": %.2f, \"sphere_mix_tflops\": %.2f, \"spec_tflops\": 157.3}\n",
           best_scalar * 1e-12, best_packed * 1e-12, best_mix * 1e-12);
    return 0;
}
