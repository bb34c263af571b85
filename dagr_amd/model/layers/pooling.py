"""Mirror of ``src/dagr/model/layers/pooling.py:19-97`` (voxel-grid pooling).  Buffers are
non-persistent exactly like the reference's, so they do not appear in the state_dict."""
import torch


class Pooling(torch.nn.Module):
    def __init__(self, size, width, height, batch_size, transform, aggr="max", keep_temporal_ordering=False,
                 dim=2, self_loop=False, in_channels=-1):
        super().__init__()
        assert aggr in ["mean", "max"]
        assert not self_loop and in_channels <= 0, \
            "only the Pooling options reachable from Net (net.py:78-97) are implemented"
        self.aggr = aggr
        # --keep_temporal_ordering (pooling.py:69-72): honoured by this module's forward; the window engine's fused
        # pooling does not filter edges, so DAGR.forward takes the module path when the flag is set
        self.keep_temporal_ordering = bool(keep_temporal_ordering)
        self.register_buffer("voxel_size", torch.cat([size, torch.Tensor([1])]), persistent=False)
        self.transform = transform
        self.dim = dim
        self.register_buffer("start", torch.Tensor([0, 0, 0, 0]), persistent=False)
        self.register_buffer("end", torch.Tensor([0.9999999, 0.9999999, 0.9999999, batch_size - 1]), persistent=False)
        self.register_buffer("wh_inv", 1 / torch.Tensor([[width, height]]), persistent=False)
        self.batch_size = batch_size
        self.max_num_voxels = batch_size * self.num_grid_cells

    def forward(self, data):
        """``pooling.py:51-97``: voxel clusters, max / mean features, mean position floored to the pixel grid, the unique
        coarse edges without self loops and their Cartesian attributes."""
        from . import _ops
        return _ops.voxel_pool(self, data)

    @property
    def num_grid_cells(self):
        return int((1 / self.voxel_size + 1e-3).int().prod())
