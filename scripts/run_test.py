#!/usr/bin/env python
"""Script-level twin of the reference's ``scripts/run_test.py`` (:31-66) / ``run_test_interframe.py`` (:21-45) for this
stack: same flow -- flags, dataset, loader, ``DAGR(args, height, width).cuda()``, ``ModelEMA``, checkpoint into
``ema.ema`` (strict), ``cache_luts``, no-grad loop ``detections, targets = model(format_data(batch))`` -- with the
window batches sharded over the ranks of a ``torch.distributed.run`` launch (one process per GPU, no collective on
the data path) and one RCCL gather of the detections at the end (``dagr_amd/parallel.py``).

The DSEC / N-Caltech101 readers are outside the hot path (their h5 / blosc
dependencies are absent), so the dataset here is the synthetic event stream of the benchmark
(``dagr_amd/utils/synthetic.py``, the reference's sample contract: ``data/utils.py:6-19``, ``dsec_data.py:141-147``).
Without ``--checkpoint`` the model keeps seeded random weights.  Output: ``<output_directory>/detections.npy`` with the
record layout of ``utils/buffers.py:46-66`` plus the window id.

  python scripts/run_test.py --config dagr-s --windows 64 --batch_size 8 --output_directory /tmp/out
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_test.py ...
"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagr_amd import parallel                                       # noqa: E402
from dagr_amd.data import Batch, Data                               # noqa: E402
from dagr_amd.model.networks.dagr import DAGR                       # noqa: E402
from dagr_amd.model.networks.ema import ModelEMA                    # noqa: E402
from dagr_amd.utils import synthetic as syn                         # noqa: E402
from dagr_amd.utils.args import MODEL_CONFIGS, model_args           # noqa: E402
from dagr_amd.utils.buffers import format_data                      # noqa: E402
from dagr_amd.utils.testing_weights import randomize_               # noqa: E402

RECORD = [("window", "<u4"), ("t", "<u8"), ("x", "<f4"), ("y", "<f4"), ("w", "<f4"), ("h", "<f4"), ("class_id", "u1"),
          ("class_confidence", "<f4")]


class SyntheticWindows:
    """``len`` windows of ``n_events`` events each, ``__getitem__`` -> the reference's per-sample ``Data``."""

    def __init__(self, n_windows, n_events, width, height, stream="uniform", use_image=False, seed=1234):
        self.n, self.n_events, self.width, self.height = n_windows, n_events, width, height
        self.gen = syn.uniform_window if stream == "uniform" else syn.edges_window
        self.use_image, self.seed = use_image, seed

    def __len__(self):
        return self.n

    def __getitem__(self, w):
        x, y, t, p = self.gen(self.n_events, self.width, self.height, seed=self.seed + w)
        d = Data(x=torch.from_numpy(p.reshape(-1, 1)), pos=torch.from_numpy(np.stack([x, y], -1)), t=torch.from_numpy(t),
                 width=self.width, height=self.height, time_window=1000000, sequence=f"synthetic{w // 100:03d}", window=w)
        if self.use_image:
            d.image = torch.randint(0, 256, (1, 3, self.height, self.width), dtype=torch.uint8,
                                    generator=torch.Generator().manual_seed(self.seed + w))
        return d


def parse(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--config", default="dagr-s", choices=sorted(MODEL_CONFIGS))
    p.add_argument("--checkpoint", type=Path, default=None, help="torch.load(path)['ema'] -> ema.ema (strict)")
    p.add_argument("--output_directory", type=Path, default=Path("run_test_out"))
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--windows", type=int, default=32)
    p.add_argument("--events_per_window", type=int, default=50000)
    p.add_argument("--width", type=int, default=640)
    p.add_argument("--height", type=int, default=480)
    p.add_argument("--stream", default="uniform", choices=["uniform", "edges"])
    p.add_argument("--use_image", action="store_true")
    p.add_argument("--img_net", default="resnet50")
    return p.parse_args(argv)


def main(argv=None):
    a = parse(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.manual_seed(42)
    np.random.seed(42)
    torch.cuda.set_device(local_rank)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    args = model_args(a.config, batch_size=a.batch_size, use_image=a.use_image, img_net=a.img_net)
    dataset = SyntheticWindows(a.windows, a.events_per_window, a.width, a.height, a.stream, a.use_image)
    # window batches of this rank: batch k holds windows [k*B, (k+1)*B); batch k -> rank k mod G (drop_last=True)
    n_batches = len(dataset) // a.batch_size
    my_batches = parallel.shard_indices(n_batches, rank, world)

    model = DAGR(args, height=a.height, width=a.width)
    if a.checkpoint is None:
        model = randomize_(model, seed=0)
    model = model.cuda()
    ema = ModelEMA(model)
    if a.checkpoint is not None:
        ema.ema.load_state_dict(torch.load(a.checkpoint, map_location="cuda")["ema"])
    else:
        ema.ema.load_state_dict(model.state_dict())
    ema.ema.cache_luts(radius=args.radius, height=a.height, width=a.width)
    net = ema.ema.eval()

    rows = []
    n_events = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for k in my_batches:
            samples = [dataset[k * a.batch_size + i] for i in range(a.batch_size)]
            batch = Batch.from_data_list(samples).cuda()
            n_events += int(batch.pos.shape[0])
            windows = [s.window for s in samples]
            detections = net(format_data(batch), return_targets=False)[0]
            for w, det in zip(windows, detections):
                n = det["boxes"].shape[0]
                if n:
                    rows.append(torch.cat([torch.full((n, 1), float(w), device=det["boxes"].device), det["boxes"],
                                           det["scores"].view(-1, 1), det["labels"].float().view(-1, 1)], 1))
    mine = torch.cat(rows, 0) if rows else torch.zeros((0, 7), device="cuda")
    allrows = parallel.restore_window_order(parallel.gather_detections(mine))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tot = torch.tensor([float(n_events)], device="cuda")
        torch.distributed.all_reduce(tot)
        n_events = int(tot.item())
    if rank == 0:
        r = allrows.cpu().numpy()
        rec = np.zeros((len(r),), dtype=RECORD)
        rec["window"] = r[:, 0]
        rec["t"] = 1000000                                   # every window ends at time_window (dsec_data.py:145)
        rec["x"], rec["y"] = r[:, 1], r[:, 2]
        rec["w"], rec["h"] = r[:, 3] - r[:, 1], r[:, 4] - r[:, 2]
        rec["class_confidence"], rec["class_id"] = r[:, 5], r[:, 6]
        a.output_directory.mkdir(parents=True, exist_ok=True)
        np.save(a.output_directory / "detections.npy", rec)
        print(f"{n_batches * a.batch_size} windows, {n_events} events, {len(rec)} detections on {world} GPU(s) in {dt:.2f} s "
              f"(incl. synthetic data generation on the host) -> {a.output_directory / 'detections.npy'}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
