"""``asy_tools`` with the reference's function names, argument order and in-place conventions
(``src/dagr/asynchronous/asy_tools/main.cu:239-244``; call sites ``asynchronous/linear.py``, ``batch_norm.py``,
``conv.py``, ``max_pool.py``) over libdagr_hip.  Tensors stay the caller's; ``RuntimeError`` for host / strided inputs
as the reference's ``AT_ASSERTM`` checks raise (main.cu:8-11)."""
from .. import _lib


def _check(**tensors):
    dev = None
    for name, t in tensors.items():
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")
        if dev is not None and t.device != dev:
            raise RuntimeError(f"{name} must be on the same device as the other arguments")
        dev = t.device
    return dev


def masked_lin(indices, x_in, x_out, weight, bias, add):
    """x_out[i] (+)= x_in[i] @ weight.T + bias for i in indices (main.cu:220-236)."""
    dev = _check(indices=indices, x_in=x_in, x_out=x_out, weight=weight, bias=bias)
    cout, cin = weight.shape
    P = _lib.ptr
    _lib.check(_lib.lib().dagr_masked_lin(P(indices), P(x_in), P(x_out), P(weight), P(bias), 1 if add else 0,
                                          indices.shape[0], cin, cout, _lib.cur_stream(dev)), "masked_lin")


def masked_lin_no_bias(indices, x_in, x_out, weight, add):
    """x_out[i] (+)= x_in[i] @ weight.T for i in indices (main.cu:198-216)."""
    dev = _check(indices=indices, x_in=x_in, x_out=x_out, weight=weight)
    cout, cin = weight.shape
    P = _lib.ptr
    _lib.check(_lib.lib().dagr_masked_lin_no_bias(P(indices), P(x_in), P(x_out), P(weight), 1 if add else 0,
                                                  indices.shape[0], cin, cout, _lib.cur_stream(dev)), "masked_lin_no_bias")


def masked_isdiff(indices, x_old, x_new, atol, rtol):
    """The subset of ``indices`` whose rows differ: any |old - new| > atol + rtol * new.  Like the reference, marks the
    others with -1 in ``indices`` itself and returns the compacted survivors (main.cu:112-139)."""
    dev = _check(indices=indices, x_old=x_old, x_new=x_new)
    P = _lib.ptr
    _lib.check(_lib.lib().dagr_masked_isdiff(P(indices), P(x_old), P(x_new), float(atol), float(rtol), indices.shape[0],
                                             x_old.shape[1], _lib.cur_stream(dev)), "masked_isdiff")
    return indices[indices > -1]


def masked_inplace_BN(indices, x, x_out, running_mean, running_var, weight, bias, eps):
    """x_out[i] = (x[i] - mean) / sqrt(var + eps) * weight + bias for i in indices (main.cu:69-96)."""
    dev = _check(indices=indices, x=x, x_out=x_out, running_mean=running_mean, running_var=running_var, weight=weight,
                 bias=bias)
    P = _lib.ptr
    _lib.check(_lib.lib().dagr_masked_inplace_BN(P(indices), P(x), P(x_out), P(running_mean), P(running_var), P(weight),
                                                 P(bias), float(eps), indices.shape[0], x.shape[1],
                                                 _lib.cur_stream(dev)), "masked_inplace_BN")
