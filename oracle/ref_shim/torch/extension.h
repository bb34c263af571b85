// oracle/ref_shim/torch/extension.h -- TEST INFRASTRUCTURE ONLY.
// Minimal stand-in for <torch/extension.h> so that the reference's ev_graph.cu and asy_tools/main.cu can be compiled
// UNMODIFIED, from /root/reference, by hipcc (oracle/Makefile target `ref`).  It provides exactly
// the surface that translation unit touches: a non-owning tensor view over a raw device pointer
// (type().is_cuda(), is_contiguous(), device().index(), size(i), data<T>()), AT_ASSERTM and a
// do-nothing PYBIND11_MODULE.  This is how the oracle is pinned to the reference's real kernels;
// it is never part of the product build.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define AT_ASSERTM(cond, msg)                                              \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "ref assert failed: %s\n", msg);               \
            abort();                                                       \
        }                                                                  \
    } while (0)

namespace torch {
struct ShimType { bool is_cuda() const { return true; } };
struct ShimDevice { int index() const { return 0; } };
struct ShimMask {};   // `indices > -1` of asy_tools/main.cu:138: the compaction itself is done by the test harness
struct Tensor {
    void *ptr = nullptr;
    std::vector<int64_t> shape;
    ShimType type() const { return {}; }
    bool is_contiguous() const { return true; }
    ShimDevice device() const { return {}; }
    int64_t size(int i) const { return shape[i]; }
    template <typename T> T *data() const { return reinterpret_cast<T *>(ptr); }
    Tensor index(std::initializer_list<ShimMask>) const { return *this; }
};
inline ShimMask operator>(const Tensor &, int) { return {}; }
}  // namespace torch

struct ShimModule {
    template <typename F> void def(const char *, F, const char *) {}
};
#define TORCH_EXTENSION_NAME shim
#define PYBIND11_MODULE(name, m) [[maybe_unused]] static void dagr_ref_shim_module(ShimModule &m)
