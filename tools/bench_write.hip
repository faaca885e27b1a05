// Write-rate probe for the scatter pass: how fast can workgroups stream out contiguous per-workgroup regions?
//   hipcc -O3 --offload-arch=gfx950 tools/bench_write.hip -o tools/bench_write && tools/bench_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long ull2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// every workgroup writes `per_wg` 16-byte slots, contiguous, in pieces of 256 slots; SYNC: two barriers per `round` pieces
template <bool NT, bool SYNC>
__global__ __launch_bounds__(256) void regions(ull2v *out, unsigned per_wg, unsigned round, unsigned shift) {
    __shared__ ull2v stage[2048];
    ull2v *dst = out + (size_t)blockIdx.x * per_wg + shift;
    for (unsigned base = 0; base < per_wg; base += round * 256u) {
        if (SYNC) {
            for (unsigned i = threadIdx.x; i < 2048u; i += 256u) stage[i] = ull2v{base + i, blockIdx.x};
            __syncthreads();
        }
        for (unsigned p = 0; p < round; p++) {
            const unsigned i = base + p * 256u + threadIdx.x;
            if (i < per_wg) {
                const ull2v v = SYNC ? stage[(p * 256u + threadIdx.x) & 2047u] : ull2v{i, blockIdx.x};
                if (NT) __builtin_nontemporal_store(v, dst + i);
                else dst[i] = v;
            }
        }
        if (SYNC) __syncthreads();
    }
}

template <typename F>
static float timeit(F f, int reps = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e3f;
}

int main() {
    for (size_t total_mb : {411, 4230}) {
        const unsigned per_wg = 4224;  // 16-byte slots: 67.6 KB per workgroup, what a half-visible tile writes
        const unsigned wgs = (unsigned)(total_mb * 1000000ull / (per_wg * 16ull));
        ull2v *buf;
        CK(hipMalloc(&buf, (size_t)(wgs + 1) * per_wg * 16));
        printf("%zu MB in %u workgroups of %u bytes\n", total_mb, wgs, per_wg * 16);
        for (unsigned shift : {0u, 3u}) {
            float t;
            t = timeit([&] { hipLaunchKernelGGL((regions<true, false>), dim3(wgs), dim3(256), 0, 0, buf, per_wg, 1u, shift); });
            printf("  shift %u  nt, no staging              %8.1f us  %5.2f TB/s\n", shift, t, wgs * (double)per_wg * 16 / t / 1e6);
            t = timeit([&] { hipLaunchKernelGGL((regions<false, false>), dim3(wgs), dim3(256), 0, 0, buf, per_wg, 1u, shift); });
            printf("  shift %u  plain, no staging           %8.1f us  %5.2f TB/s\n", shift, t, wgs * (double)per_wg * 16 / t / 1e6);
            t = timeit([&] { hipLaunchKernelGGL((regions<true, true>), dim3(wgs), dim3(256), 0, 0, buf, per_wg, 4u, shift); });
            printf("  shift %u  nt, staged rounds of 16 KB  %8.1f us  %5.2f TB/s\n", shift, t, wgs * (double)per_wg * 16 / t / 1e6);
            t = timeit([&] { hipLaunchKernelGGL((regions<true, true>), dim3(wgs), dim3(256), 0, 0, buf, per_wg, 8u, shift); });
            printf("  shift %u  nt, staged rounds of 32 KB  %8.1f us  %5.2f TB/s\n", shift, t, wgs * (double)per_wg * 16 / t / 1e6);
        }
        CK(hipFree(buf));
    }
    return 0;
}
