// Measures how fast a CU serves per-lane 16-byte gathers whose 64 addresses fall into 64
// different cache lines -- the access pattern of one step of the per-lane hierarchy traversal
// (RT_FETCH_BVH) -- from tables that fit the vector L1 (16 KB), the L2 (128 KB, 1 MB) and, for
// comparison, the same pattern served from LDS.  Each lane follows its own dependent chain
// index -> table[index].w, as a traversal follows its links.
//   usage: gather_peak       (one line per table size and occupancy)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int STEPS = 2048;

__global__ __launch_bounds__(256) void chase_global(const uint4 *__restrict__ table, uint32_t mask, uint32_t *out) {
    uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
    uint32_t acc = 0;
    for (int s = 0; s < STEPS; ++s) {
        const uint4 v = table[i];
        acc += v.x ^ v.y ^ v.z;
        i = v.w & mask;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void chase_lds(const uint4 *__restrict__ table, uint32_t n, uint32_t *out) {
    extern __shared__ uint4 lds[];
    for (uint32_t k = threadIdx.x; k < n; k += 256) lds[k] = table[k];
    __syncthreads();
    const uint32_t mask = n - 1;
    uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
    uint32_t acc = 0;
    for (int s = 0; s < STEPS; ++s) {
        const uint4 v = lds[i];
        acc += v.x ^ v.y ^ v.z;
        i = v.w & mask;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("device   CUs %d  clock %.2f GHz\n", cus, ghz);
    uint32_t *out;
    CHECK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (uint32_t n : {1024u, 8192u, 65536u, 1u << 20}) { // entries of 16 bytes
        std::vector<uint4> h(n);
        uint64_t x = 88172645463325252ull;
        for (uint32_t k = 0; k < n; ++k) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            h[k] = {static_cast<uint32_t>(x), static_cast<uint32_t>(x >> 11), k, static_cast<uint32_t>(x >> 32)};
        }
        uint4 *d;
        CHECK(hipMalloc(&d, n * sizeof(uint4)));
        CHECK(hipMemcpy(d, h.data(), n * sizeof(uint4), hipMemcpyHostToDevice));
        for (int waves_per_simd : {2, 4, 8}) {
            const int blocks = cus * waves_per_simd; // 256 threads = 4 waves = one per SIMD
            for (int variant = 0; variant < 2; ++variant) {
                if (variant == 1 && n * sizeof(uint4) > 16384) continue; // 16 KB per block keeps 8 blocks per CU resident
                for (int rep = 0; rep < 2; ++rep) {
                    CHECK(hipEventRecord(e0));
                    if (variant == 0) hipLaunchKernelGGL(chase_global, dim3(blocks), dim3(256), 0, 0, d, n - 1, out);
                    else hipLaunchKernelGGL(chase_lds, dim3(blocks), dim3(256), n * sizeof(uint4), 0, d, n, out);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                }
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double lookups = double(blocks) * 256.0 * STEPS;
                printf("%-6s table %7.0f KB  waves/SIMD %d: %7.1f G lane-gathers/s  = %.2f per CU per clock (%.1f cycles per wave64 gather per CU)\n",
                       variant ? "LDS" : "global", n * 16.0 / 1024, waves_per_simd, lookups / ms * 1e-6, lookups / (ms * 1e-3) / cus / (ghz * 1e9),
                       64.0 / (lookups / (ms * 1e-3) / cus / (ghz * 1e9)));
            }
        }
        CHECK(hipFree(d));
    }
    return 0;
}
