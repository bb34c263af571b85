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
    def __init__(self, args, height, width):
        super().__init__()
        channels = [1, int(args.base_width * 32), int(args.after_pool_width * 64), int(args.net_stem_width * 128),
                    int(args.net_stem_width * 128), int(args.net_stem_width * 128)]
        self.height, self.width = height, width
        self.out_channels_cnn = []
        self.use_image = bool(args.use_image)
        if self.use_image:
            from .net_img import HookModule, make_img_net
            self.out_channels_cnn = [256, 256]
            self.net = HookModule(make_img_net(args.img_net), input_channels=3, height=height, width=width,
                                  feature_layers=["conv1", "layer1", "layer2", "layer3", "layer4"],
                                  output_layers=["layer3", "layer4"], feature_channels=channels[1:],
                                  output_channels=self.out_channels_cnn)
        self.num_scales = args.num_scales
        self.num_classes = dict(dsec=2, ncaltech101=100).get(args.dataset, 2)
        self.events_to_graph = EV_TGN(args)

        output_channels = channels[1:]
        self.out_channels = output_channels[-2:]
        input_channels = channels[:-1]
        if self.use_image:
            input_channels = [input_channels[i] + self.net.feature_channels[i] for i in range(len(input_channels))]
        self.input_channels = input_channels
        self.output_channels = output_channels

        poolings = compute_pooling_at_each_layer(args.pooling_dim_at_output, num_layers=4)
        max_vals_for_cartesian = 2 * poolings[:, :2].max(-1).values
        self.strides = torch.ceil(poolings[-2:, 1] * height).numpy().astype("int32").tolist()
        self.strides = self.strides[-self.num_scales:]

        effective_radius = 2 * float(int(args.radius * width + 2) / width)
        self.edge_attrs = Cartesian(norm=True, cat=False, max_value=effective_radius)
        self.conv_block1 = Layer(2 + input_channels[0], output_channels[0], args=args)
        cart1 = Cartesian(norm=True, cat=False, max_value=2 * effective_radius)
        self.pool1 = Pooling(poolings[0], width=width, height=height, batch_size=args.batch_size, transform=cart1,
                             aggr=args.pooling_aggr, keep_temporal_ordering=args.keep_temporal_ordering)
        self.layer2 = Layer(input_channels[1] + 2, output_channels[1], args=args)
        cart2 = Cartesian(norm=True, cat=False, max_value=max_vals_for_cartesian[1])
        self.pool2 = Pooling(poolings[1], width=width, height=height, batch_size=args.batch_size, transform=cart2,
                             aggr=args.pooling_aggr, keep_temporal_ordering=args.keep_temporal_ordering)
        self.layer3 = Layer(input_channels[2] + 2, output_channels[2], args=args)
        cart3 = Cartesian(norm=True, cat=False, max_value=max_vals_for_cartesian[2])
        self.pool3 = Pooling(poolings[2], width=width, height=height, batch_size=args.batch_size, transform=cart3,
                             aggr=args.pooling_aggr, keep_temporal_ordering=args.keep_temporal_ordering)
        self.layer4 = Layer(input_channels[3] + 2, output_channels[3], args=args)
        cart4 = Cartesian(norm=True, cat=False, max_value=max_vals_for_cartesian[3])
        self.pool4 = Pooling(poolings[3], width=width, height=height, batch_size=args.batch_size, transform=cart4,
                             aggr="mean", keep_temporal_ordering=args.keep_temporal_ordering)
        self.layer5 = Layer(input_channels[4] + 2, output_channels[4], args=args)

    def get_output_sizes(self):  # net.py:103-106
        poolings = [self.pool3.voxel_size[:2], self.pool4.voxel_size[:2]]
        return [(1 / p + 1e-3).cpu().int().numpy().tolist()[::-1] for p in poolings]
