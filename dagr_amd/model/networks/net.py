"""Mirror of ``src/dagr/model/networks/net.py`` (Net :31-190): same sub-module names / state_dict
keys (``conv_block1``, ``pool1``, ``layer2`` ... ``layer5``, ``net`` for the image branch) and the same
derived constants (poolings, strides, cartesian maxima -- built with torch fp32 ops exactly as the
reference does, SURVEY QUIRK-11).  ``forward`` runs through ``dagr_amd.engine``."""
import torch

from ..layers.components import Cartesian
from ..layers.conv import Layer
from ..layers.ev_tgn import EV_TGN
from ..layers.pooling import Pooling


def compute_pooling_at_each_layer(pooling_dim_at_output, num_layers):  # net.py:19-28
    py, px = map(int, pooling_dim_at_output.split("x"))
    pooling_base = torch.tensor([1.0 / px, 1.0 / py, 1.0 / 1])
    poolings = []
    for i in range(num_layers):
        pooling = pooling_base / 2 ** (3 - i)
        pooling[-1] = 1
        poolings.append(pooling)
    return torch.stack(poolings)


class Net(torch.nn.Module):
    """Parameter holder + derived constants of the backbone.  Sub-module names (``conv_block1``, ``pool1``,
    ``layer2`` ... ``layer5``, ``net``) are the reference's, so checkpoints map one to one."""

    LAYER_NAMES = ("conv_block1", "layer2", "layer3", "layer4", "layer5")

    def __init__(self, args, height, width):
        super().__init__()
        self.height, self.width = height, width
        stem = int(args.net_stem_width * 128)
        widths = [1, int(args.base_width * 32), int(args.after_pool_width * 64), stem, stem, stem]   # net.py:35-38
        self.use_image = bool(args.use_image)
        self.num_scales = args.num_scales
        self.num_classes = {"dsec": 2, "ncaltech101": 100}.get(args.dataset, 2)
        self.out_channels_cnn = []
        if self.use_image:
            from .net_img import HookModule, make_img_net
            self.out_channels_cnn = [256, 256]
            self.net = HookModule(make_img_net(args.img_net), input_channels=3, height=height, width=width,
                                  feature_layers=["conv1", "layer1", "layer2", "layer3", "layer4"],
                                  output_layers=["layer3", "layer4"], feature_channels=widths[1:],
                                  output_channels=self.out_channels_cnn)
        self.events_to_graph = EV_TGN(args)

        self.output_channels = widths[1:]
        self.out_channels = self.output_channels[-2:]
        extra = self.net.feature_channels if self.use_image else [0] * 5
        self.input_channels = [c + e for c, e in zip(widths[:-1], extra)]

        # voxel sizes, head strides and cartesian maxima exactly as the reference derives them -- with torch
        # fp32 ops, not python floats (SURVEY QUIRK-11)
        poolings = compute_pooling_at_each_layer(args.pooling_dim_at_output, num_layers=4)
        cart_max = 2 * poolings[:, :2].max(-1).values
        self.strides = torch.ceil(poolings[-2:, 1] * height).numpy().astype("int32").tolist()[-self.num_scales:]
        r_eff = 2 * float(int(args.radius * width + 2) / width)
        self.edge_attrs = Cartesian(norm=True, cat=False, max_value=r_eff)
        pool_max = [2 * r_eff, cart_max[1], cart_max[2], cart_max[3]]                    # net.py:77,83,89,95
        pool_aggr = [args.pooling_aggr] * 3 + ["mean"]                                    # net.py:96-97
        for k, name in enumerate(self.LAYER_NAMES):
            setattr(self, name, Layer(self.input_channels[k] + 2, self.output_channels[k], args=args))
            if k < 4:
                setattr(self, f"pool{k + 1}",
                        Pooling(poolings[k], width=width, height=height, batch_size=args.batch_size,
                                transform=Cartesian(norm=True, cat=False, max_value=pool_max[k]), aggr=pool_aggr[k],
                                keep_temporal_ordering=args.keep_temporal_ordering))

    def forward(self, data, reset=True):
        """``Net.forward`` (net.py:108-190), module by module (eval mode): graph -> [image features] -> Cartesian edge
        attributes -> Layer / Pooling x 4; returns ``[out3, out4][-num_scales:]`` (+ the image outputs).  Whole windows
        are faster through ``DAGR.forward`` / the engine; this is the same computation as separate operators."""
        from ..layers import _ops
        from ..utils import shallow_copy
        if self.use_image:
            image_feat, image_outputs = self.net(data.image)
        if hasattr(data, "reset"):
            reset = data.reset
        data = self.events_to_graph(data, reset=reset)

        def with_image(d, k):
            return torch.cat((d.x, _ops.sample_features(d, image_feat[k].detach(), self.width, self.height)), dim=1)
        if self.use_image:
            data.x = with_image(data, 0)
        data = self.edge_attrs(data)
        if getattr(data, "is_lazy", None) is not None and data.is_lazy("edge_attr"):
            recipe = data.__dict__["_lazy"]["edge_attr"]
            data.set_lazy("edge_attr", lambda d, f=recipe: torch.clamp(f(d), min=0, max=1))
            # the clamp never binds while every edge's pixel offset lies inside the Cartesian range (|d| <= search radius
            # <= max * size on both axes: r_eff above); should a configuration break that, the convs go by the attributes
            rpx = int(self.events_to_graph.radius * self.width + 1)
            if not (rpx <= self.edge_attrs.max * self.width and rpx <= self.edge_attrs.max * self.height):
                data._dagr_pixel_codes = None
        else:
            data.edge_attr = torch.clamp(data.edge_attr, min=0, max=1)
        outputs = []
        for k, name in enumerate(self.LAYER_NAMES):
            data.x = torch.cat((data.x, data.pos[:, :2]), dim=1)
            data = getattr(self, name)(data)
            if k == 3:
                out3 = shallow_copy(data)
                out3.pooling = self.pool3.voxel_size[:3]
                outputs.append(out3)
            if k == 4:
                data.pooling = self.pool4.voxel_size[:3]
                outputs.append(data)
                break
            if self.use_image:
                data.x = with_image(data, k + 1)
            data = getattr(self, f"pool{k + 1}")(data)
        if self.use_image:
            return outputs[-self.num_scales:], image_outputs[-self.num_scales:]
        return outputs[-self.num_scales:]

    def get_output_sizes(self):  # net.py:103-106
        key = tuple((p.voxel_size.data_ptr(), p.voxel_size._version) for p in (self.pool3, self.pool4))
        cached = getattr(self, "_output_sizes", None)
        if cached is None or cached[0] != key:      # (a read-back per call would synchronise every training step)
            cached = self._output_sizes = (key, [(1 / p.voxel_size[:2] + 1e-3).cpu().int().numpy().tolist()[::-1]
                                                 for p in (self.pool3, self.pool4)])
        return [list(v) for v in cached[1]]
