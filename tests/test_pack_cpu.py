"""Host-side weight packing of the fused pooled-level conv (dagr_amd/engine.py:_ConvPack): the MFMA operand
order documented in include/dagr_hip.h, checked element by element on CPU tensors."""
import numpy as np
import torch

from dagr_amd.engine import _ConvPack


def test_wq_is_the_documented_mfma_operand_order():
    rng = np.random.default_rng(3)
    for cin, cskip, N in ((3, 0, 16), (5, 2, 21), (32, 0, 64), (64, 32, 128)):
        K = 26 * cin + cskip
        Wm = torch.from_numpy(rng.standard_normal((K, N)).astype(np.float32))
        pack = _ConvPack(cin, cskip, Wm, torch.zeros(N), relu=True)
        assert pack.K == K and pack.N == N and pack.ldw % 8 == 0 and pack.ldw >= N
        assert torch.equal(pack.Wm[:, :N], Wm) and (pack.Wm[:, N:] == 0).all()
        C, G = (N + 15) // 16, (K + 15) // 16
        Wq = pack.Wq
        assert tuple(Wq.shape) == (C, G, 4, 16, 4) and Wq.is_contiguous()
        flat = Wq.reshape(C, G, 64, 4)     # lane l = 16 * (l >> 4) + (l & 15)
        for c, g, l, j in ((0, 0, 0, 0), (C - 1, G - 1, 63, 3), (C // 2, G // 2, 17, 2), (0, G - 1, 48, 1)):
            k, n = 16 * g + 4 * j + (l >> 4), 16 * c + (l & 15)
            want = float(Wm[k, n]) if (k < K and n < N) else 0.0
            assert float(flat[c, g, l, j]) == want
        # every weight appears exactly once, the rest is zero padding
        assert abs(float(Wq.abs().sum()) - float(Wm.abs().sum())) < 1e-3 * float(Wm.abs().sum())
