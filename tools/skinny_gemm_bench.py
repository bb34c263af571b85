#!/usr/bin/env python
"""tools/skinny_gemm_bench.py -- the three library GEMMs of a level-0 SplineConv's training step are skinny (400 k rows, 16
outputs): time torch.mm against re-shaped forms (split-K as a batched product; row blocks as a batch).  Builder tool."""
import sys
import torch

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for K, cout in ((416, 16), (78, 16), (468, 64), (1682, 64)):
    nn = n if K <= 416 else 17922
    A = torch.randn(nn, K, device=dev)
    g = torch.randn(nn, cout, device=dev)
    Wm = torch.randn(K, cout, device=dev)
    out = {}
    out["A^T g: mm"] = timed(lambda: A.t() @ g)
    ref = A.t() @ g
    for P in (16, 64, 256):
        m = nn // P * P

        def splitk(P=P, m=m):
            r = torch.bmm(A[:m].view(P, m // P, K).transpose(1, 2), g[:m].view(P, m // P, cout)).sum(0)
            if m < nn:
                r = r + A[m:].t() @ g[m:]
            return r
        out[f"A^T g: split-K bmm P={P}"] = timed(splitk)
        err = float((splitk() - ref).abs().max() / ref.abs().max())
        out[f"  rel diff P={P}"] = err
    out["g Wm^T: mm"] = timed(lambda: g @ Wm.t())
    WmT = Wm.t().contiguous()
    out["g Wm^T: mm (Wm^T contiguous)"] = timed(lambda: g @ WmT)
    for P in (64, 1024):
        m = nn // P * P
        out[f"g Wm^T: matmul over {P} row blocks"] = timed(lambda P=P, m=m: torch.matmul(g[:m].view(P, m // P, cout), WmT))
    out["g Wm^T: addmm into preallocated"] = timed(lambda: torch.mm(g, WmT, out=torch.empty(nn, K, device=dev)))
    out["A Wm: mm"] = timed(lambda: A @ Wm)
    for P in (64, 1024):
        m = nn // P * P
        out[f"A Wm: matmul over {P} row blocks"] = timed(lambda P=P, m=m: torch.matmul(A[:m].view(P, m // P, K), Wm))
    print(f"n={nn} K={K} cout={cout}")
    for k, v in out.items():
        print(f"   {k:45s} {v:10.1f}" + (" us" if not k.startswith("  rel") else ""))
