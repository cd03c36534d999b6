// What does HBM3E give a gather of RANDOM, ALIGNED records of S bytes?  The ceiling the HJI lookup layouts sit under: one lookup = 4096 B of corner data as
// 1 x 4 KiB, 4 x 1 KiB or 16 x 256 B records (pg_set_hji_grid), gathered by sixteen lanes with 16-byte loads.  Here: the same access shape without any arithmetic --
// every 16-lane group reads `recs` random records of S bytes from a table far beyond the 256 MiB Infinity Cache and adds them up.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/random_read_probe.hip -o tools/probes/random_read_probe ;  run on the GPU box: tools/probes/random_read_probe [table GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int S>      // record bytes: 256, 512, 1024, 4096
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ table, size_t n_rec, int recs_per_group, unsigned long long seed, float4* __restrict__ out) {
    const int lane = threadIdx.x & 15;
    const size_t grp = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long x = seed + grp * 0x9E3779B97F4A7C15ull;
    for (int r = 0; r < recs_per_group; r++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;                     // xorshift64*: the same record index in the sixteen lanes of the group
        const size_t rec = (size_t)((x * 0x2545F4914F6CDD1Dull) >> 11) % n_rec;
        const float4* p = table + rec * (S / 16) + lane;
#pragma unroll
        for (int k = 0; k < S / 256; k++) { const float4 v = p[16 * k]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const size_t bytes = (size_t)(gib * (1ull << 30));
    float4* table = nullptr; float4* out = nullptr;
    CK(hipMalloc((void**)&table, bytes)); CK(hipMemset(table, 0, bytes));
    const int groups = 1 << 20;                                       // 2^20 "lookups" of 4096 B each, like the bench
    CK(hipMalloc((void**)&out, (size_t)groups * 16 * sizeof(float4)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("table %.1f GiB, %d groups of 16 lanes, 4096 B per group and launch\n", gib, groups);
    auto run = [&](auto kern, int S) -> int {
        const int recs = 4096 / S; const size_t n_rec = bytes / S;
        const dim3 grid(groups * 16 / 256), block(256);
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kern, grid, block, 0, 0, table, n_rec, recs, 1234ull + w, out);
        CK(hipDeviceSynchronize());
        const int reps = 20;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, grid, block, 0, 0, table, n_rec, recs, 99ull + i, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        const double gbs = (double)groups * 4096 / (ms * 1e-3) / 1e9;
        printf("records of %4d B (%2d per group): %.3f ms per launch, %.0f GB/s = %.2f of 8 TB/s\n", S, recs, ms, gbs, gbs / 8000.0);
        return 0;
    };
    if (run(k_gather<4096>, 4096) || run(k_gather<1024>, 1024) || run(k_gather<512>, 512) || run(k_gather<256>, 256)) return 1;
    return 0;
}
