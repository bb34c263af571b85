"""CPU restatement (numpy) of the reference's masked row operators, ``src/dagr/asynchronous/asy_tools/main.cu``.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Pinned: the reference's own main.cu is compiled unmodified through
oracle/ref_shim into oracle/_ref/libasy_tools_ref.so (oracle/Makefile) and run on the GPU box next to these functions
and libdagr_hip's kernels (tests/test_asy_tools_gpu.py).  fp32 with one rounding per multiply and per add (numpy has no
fma): agrees with the kernels' fma chains to a few ulp, not bit for bit -- the bit-for-bit check is kernel vs kernel."""
import numpy as np


def masked_lin(indices, x_in, x_out, weight, bias, add):
    """main.cu:143-176,220-236: per selected row, per output channel, a cin-ordered accumulation, then the bias."""
    out = x_out.copy()
    for i in indices:
        for co in range(weight.shape[0]):
            acc = np.float32(out[i, co]) if add else np.float32(0)
            for ci in range(weight.shape[1]):
                acc = np.float32(acc + np.float32(x_in[i, ci] * weight[co, ci]))
            if bias is not None:
                acc = np.float32(acc + bias[co])
            out[i, co] = acc
    return out


def masked_lin_no_bias(indices, x_in, x_out, weight, add):
    """main.cu:178-216."""
    return masked_lin(indices, x_in, x_out, weight, None, add)


def masked_isdiff(indices, x_old, x_new, atol, rtol):
    """main.cu:14-40,112-139: a row survives if any |old - new| > atol + rtol * new (no abs on `new`)."""
    marked = indices.copy()
    for k, i in enumerate(indices):
        d = np.abs(x_old[i].astype(np.float32) - x_new[i].astype(np.float32))
        if not (d > np.float32(atol) + np.float32(rtol) * x_new[i]).any():
            marked[k] = -1
    return marked, marked[marked > -1]


def masked_inplace_BN(indices, x, x_out, mean, var, weight, bias, eps):
    """main.cu:42-67: (x - mean) / sqrt(var + eps) * weight + bias on the selected rows."""
    out = x_out.copy()
    for i in indices:
        t = (x[i] - mean) / np.sqrt(var + np.float32(eps), dtype=np.float32)
        out[i] = t.astype(np.float32) * weight + bias
    return out
