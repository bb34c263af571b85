// tools/microbench/sort_bench.hip -- how long does a library radix sort of a window's (pixel key, event id) pairs take?
// (decides whether "sort + index" can replace the builder's count / scan / scatter / order chain).  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    const size_t sizes[] = {25000, 131072, 800000, 1600000, 3200000};
    const int bits_list[] = {19, 22, 25};
    for (size_t n : sizes) {
        for (int bits : bits_list) {
            std::vector<unsigned> hk(n), hv(n);
            unsigned s = 12345u;
            for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; hk[i] = (s >> 7) & ((1u << bits) - 1u); hv[i] = (unsigned)i; }
            unsigned *k0, *k1, *v0, *v1;
            hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
            hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
            hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
            size_t tb = 0;
            rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, n, 0, bits, 0, false);
            void *tmp; hipMalloc(&tmp, tb);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int i = 0; i < 3; i++) rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, bits, 0, false);
            hipEventRecord(a, 0);
            const int reps = 20;
            for (int i = 0; i < reps; i++) rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, bits, 0, false);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("n %8zu bits %2d temp %8zu B  %.1f us per sort\n", n, bits, tb, 1e3f * ms / reps);
            hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
        }
    }
    return 0;
}
