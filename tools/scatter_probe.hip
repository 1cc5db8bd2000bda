// How fast can 64-byte rows leave in runs of R bytes to 16 buckets?  (what bounds k_bucket_scatter.)
// Emulates the WRITE side of the bucketed consumer only: n_slices workgroups own a contiguous slice of every bucket's
// region each and advance 16 cursors by E/16 rows per step (E = entries per tile: 256 in k_bucket_scatter), streaming
// stores, one dword per lane and instruction as there.  No reads.  hipcc --offload-arch=gfx950 -O3 -o tools/bin/scatter_probe
// tools/scatter_probe.hip ; prints GB/s per E and slice count.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_scatter_probe(uint32_t* __restrict__ out, uint64_t rows_per_bucket, uint32_t n_slices, uint32_t E, uint32_t steps, int nt) {
    // slice s of bucket k: rows [k * rows_per_bucket + s * steps * (E / 16), ... + steps * (E / 16))
    const uint32_t run = E / 16u;  // rows per bucket and step
    for (uint32_t t = 0; t < steps; ++t) {
        for (uint32_t i = threadIdx.x; i < E * 16u; i += 256u) {  // E rows x 16 dwords
            const uint32_t r = i >> 4, k = r / run, j = r - k * run;
            const uint64_t row = (uint64_t)k * rows_per_bucket + ((uint64_t)blockIdx.x * steps + t) * run + j;
            uint32_t* p = out + row * 16u + (i & 15u);
            if (nt) __builtin_nontemporal_store(i, p);
            else *p = i;
        }
    }
}

// the plain streaming ceilings for comparison: every lane one dword, or one 16-byte piece, consecutive addresses
__global__ __launch_bounds__(256) void k_fill1(uint32_t* __restrict__ out, uint64_t n, int nt) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        if (nt) __builtin_nontemporal_store((uint32_t)i, out + i);
        else out[i] = (uint32_t)i;
    }
}
__global__ __launch_bounds__(256) void k_fill4(uint4* __restrict__ out, uint64_t n, int nt) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = make_uint4((uint32_t)i, 1u, 2u, 3u);
        if (nt) __builtin_nontemporal_store(v.x, &out[i].x), __builtin_nontemporal_store(v.y, &out[i].y), __builtin_nontemporal_store(v.z, &out[i].z), __builtin_nontemporal_store(v.w, &out[i].w);
        else out[i] = v;
    }
}

int main(int argc, char** argv) {
    const uint64_t total_rows = argc > 1 ? strtoull(argv[1], 0, 10) : 41943040ull;  // 2.68 GB of rows
    uint32_t* out;
    CK(hipMalloc((void**)&out, total_rows * 64 + (1 << 20)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int width = 1; width <= 4; width += 3)
        for (int nt = 0; nt <= (width == 1 ? 1 : 0); ++nt)
            for (uint32_t grid : {2048u, 16384u, 163840u}) {
                const uint64_t n = total_rows * 16 / width;
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipEventRecord(e0));
                    if (width == 1) hipLaunchKernelGGL(k_fill1, dim3(grid), dim3(256), 0, 0, out, n, nt);
                    else hipLaunchKernelGGL(k_fill4, dim3(grid), dim3(256), 0, 0, (uint4*)out, n, nt);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                printf("plain fill, %d-byte stores%s, %6u workgroups: %.3f ms, %.2f TB/s\n", 4 * width, nt ? " (nt)" : "", grid, best,
                       (double)total_rows * 64 / best * 1e-9);
                fflush(stdout);
            }
    const uint32_t Es[] = {256, 512, 1024, 2048, 4096, 16384};
    const uint32_t slices[] = {2048, 8192, 32768};
    for (int nt = 1; nt >= 0; --nt)
        for (uint32_t ns : slices)
            for (uint32_t E : Es) {
                const uint64_t rows_per_slice_bucket = total_rows / 16 / ns;  // rows of one bucket in one slice
                const uint32_t steps = (uint32_t)(rows_per_slice_bucket / (E / 16));
                if (steps == 0) continue;
                const uint64_t rows_per_bucket = (uint64_t)ns * steps * (E / 16);
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_scatter_probe, dim3(ns), dim3(256), 0, 0, out, rows_per_bucket, ns, E, steps, nt);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                const double bytes = (double)rows_per_bucket * 16 * 64;
                printf("%s stores, %5u slices, tile of %5u entries (runs of %6u bytes), %4u steps: %.3f ms, %.2f TB/s\n", nt ? "nt   " : "plain", ns, E,
                       E / 16 * 64, steps, best, bytes / best * 1e-9);
                fflush(stdout);
            }
    return 0;
}
