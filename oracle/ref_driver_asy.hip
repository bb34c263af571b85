// oracle/ref_driver_asy.hip -- TEST INFRASTRUCTURE ONLY.
// Pulls in the reference's own `asy_tools` translation unit (src/dagr/asynchronous/asy_tools/main.cu, path injected by
// oracle/Makefile; compiled where it lies, never copied) and exposes its four host entry points (main.cu:239-244)
// through a C ABI over raw device pointers: the reference's real masked kernels, for comparison with libdagr_hip's.
#include DAGR_REF_ASY_TOOLS_CU

static torch::Tensor view(const void *p, std::initializer_list<int64_t> shape) {
    torch::Tensor t;
    t.ptr = const_cast<void *>(p);
    t.shape = shape;
    return t;
}

extern "C" {

int ref_masked_lin(const int64_t *idx, int K, const float *x_in, float *x_out, int n_rows, const float *weight,
                   const float *bias, int Cin, int Cout, int add) {
    auto ti = view(idx, {K}), tx = view(x_in, {n_rows, Cin}), to = view(x_out, {n_rows, Cout});
    auto tw = view(weight, {Cout, Cin}), tb = view(bias, {Cout});
    masked_lin(ti, tx, to, tw, tb, add != 0);
    return (int)hipDeviceSynchronize();
}

int ref_masked_lin_no_bias(const int64_t *idx, int K, const float *x_in, float *x_out, int n_rows, const float *weight,
                           int Cin, int Cout, int add) {
    auto ti = view(idx, {K}), tx = view(x_in, {n_rows, Cin}), to = view(x_out, {n_rows, Cout});
    auto tw = view(weight, {Cout, Cin});
    masked_lin_no_bias(ti, tx, to, tw, add != 0);
    return (int)hipDeviceSynchronize();
}

// marks indices in place (-1 = row unchanged); the boolean-mask compaction of main.cu:138 is left to the caller
int ref_masked_isdiff(int64_t *idx, int K, const float *x_old, const float *x_new, int n_rows, int C, float atol,
                      float rtol) {
    auto ti = view(idx, {K}), ta = view(x_old, {n_rows, C}), tb = view(x_new, {n_rows, C});
    masked_isdiff(ti, ta, tb, atol, rtol);
    return (int)hipDeviceSynchronize();
}

int ref_masked_inplace_BN(const int64_t *idx, int K, const float *x, float *x_out, int n_rows, int C, const float *mean,
                          const float *var, const float *w, const float *b, float eps) {
    auto ti = view(idx, {K}), tx = view(x, {n_rows, C}), to = view(x_out, {n_rows, C});
    auto tm = view(mean, {C}), tv = view(var, {C}), tw = view(w, {C}), tb = view(b, {C});
    masked_inplace_BN(ti, tx, to, tm, tv, tw, tb, eps);
    return (int)hipDeviceSynchronize();
}
}
