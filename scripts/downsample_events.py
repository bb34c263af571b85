#!/usr/bin/env python
"""Twin of the reference's ``scripts/downsample_events.py`` (:126-167): stream a recording's full-resolution event file
in chunks of 100 000 events through the 2x (fx x fy) integrate-and-fire downsampler -- here the DEVICE kernel
(``dagr.data.downsample.downsample_events`` -> ``dagr_downsample_events``), the change map staying in HBM from chunk to
chunk -- and write the half-resolution file the DSEC reader expects: ``events/{x u2, y u2, p u1, t u4 relative}``,
``t_offset`` (first surviving timestamp) and the millisecond index ``ms_to_idx``.

Storage is behind two small classes: HDF5 (h5py + hdf5plugin's blosc filter, as the reference writes it; neither package
is part of this image, so that path raises with a precise message) and ``.npz`` (what the tests and this image use).

Reference behaviour kept on purpose (the output file is the contract): polarities are mapped {0, 1} -> {-1, +1} for every
full chunk but NOT for the trailing partial chunk (:160-163 -- its events integrate with p in {0, 1}), and the
polarity column is stored as written (HDF5 clamps -1 to 0 in the u1 dataset; the npz writer does the same).

  python scripts/downsample_events.py --input_path events.h5 --output_path events_2x.h5
"""
import argparse
from pathlib import Path

import numpy as np

CHUNK = 100000


def create_ms_to_idx(t_us):
    """ms_to_idx[m] = index of the first event with t >= m ms (downsample_events.py:35-41)."""
    t_ms = np.asarray(t_us) // 1000
    ms, counts = np.unique(t_ms, return_counts=True)
    table = np.zeros((int(t_ms[-1]) + 2,), dtype="uint64")
    table[ms.astype(np.int64) + 1] = counts
    return table[:-1].cumsum()


class NpzEvents:
    """Event arrays x, y, t, p in one ``.npz``."""

    def __init__(self, path):
        z = np.load(path)
        self.ev = {k: z[k] for k in ("x", "y", "t", "p")}

    def __len__(self):
        return len(self.ev["t"])

    def read(self, i0, i1):
        return {k: v[i0:i1].copy() for k, v in self.ev.items()}


class H5Events:
    """``events/{x,y,t,p}`` (+ ``t_offset``) of a DSEC ``events.h5`` (what dsec_det.io.extract_from_h5_by_index reads)."""

    def __init__(self, path):
        try:
            import hdf5plugin  # noqa: F401
            import h5py
        except ImportError as e:
            raise RuntimeError("reading .h5 event files needs h5py + hdf5plugin (not installed here); use an .npz input") from e
        self.f = h5py.File(str(path), "r")
        self.t_offset = int(self.f["t_offset"][()]) if "t_offset" in self.f else 0

    def __len__(self):
        return len(self.f["events/t"])

    def read(self, i0, i1):
        ev = {k: self.f[f"events/{k}"][i0:i1] for k in "xytp"}
        ev["t"] = ev["t"].astype(np.int64) + self.t_offset
        return ev


class EventSink:
    """Collects the surviving events; ``finish`` writes x u2 / y u2 / p u1 / t u4 (relative to the first event), t_offset
    and ms_to_idx -- to ``.npz``, or to blosc-compressed HDF5 datasets when the path ends in ``.h5``."""

    def __init__(self, path):
        self.path = Path(path)
        assert not self.path.exists(), f"{self.path} exists"
        self.parts, self.t_offset = [], None

    def add(self, events):
        if len(events["t"]) == 0:
            return
        if self.t_offset is None:
            self.t_offset = np.int64(events["t"][0])
        self.parts.append(dict(x=np.asarray(events["x"]).astype("u2"), y=np.asarray(events["y"]).astype("u2"),
                               p=np.clip(np.asarray(events["p"]).astype(np.int64), 0, 255).astype("u1"),
                               t=(np.asarray(events["t"]).astype(np.int64) - self.t_offset).astype("u4")))

    def finish(self):
        cat = {k: (np.concatenate([p[k] for p in self.parts]) if self.parts else np.zeros(0, dtype=d))
               for k, d in (("x", "u2"), ("y", "u2"), ("p", "u1"), ("t", "u4"))}
        ms_to_idx = create_ms_to_idx(cat["t"]) if len(cat["t"]) else np.zeros(0, dtype="uint64")
        t_offset = np.int64(self.t_offset if self.t_offset is not None else 0)
        if self.path.suffix == ".h5":
            try:
                import hdf5plugin  # noqa: F401
                import h5py
            except ImportError as e:
                raise RuntimeError("writing .h5 needs h5py + hdf5plugin (not installed here); use an .npz output") from e
            blosc = dict(compression=32001, compression_opts=(0, 0, 0, 0, 1, 2, 5), chunks=True)   # zstd, bit shuffle
            with h5py.File(str(self.path), "a") as f:
                for k, v in cat.items():
                    f.create_dataset(f"events/{k}", data=v, maxshape=(None,), **blosc)
                f.create_dataset("t_offset", data=t_offset, dtype="i8")
                f.create_dataset("ms_to_idx", data=ms_to_idx, dtype="u8", **blosc)
        else:
            np.savez(self.path, t_offset=t_offset, ms_to_idx=ms_to_idx, **cat)
        return {k: len(v) for k, v in cat.items()}


def device_downsampler(device="cuda"):
    """chunk dict (numpy) -> surviving chunk (numpy), the change map resident on the device between calls."""
    import torch
    from dagr.data.downsample import downsample_events
    state = {"map": None}

    def run(ev, ih, iw, oh, ow):
        dev = {k: torch.from_numpy(np.ascontiguousarray(v.astype(np.int64) if k != "p" else v.astype(np.int8))).to(device)
               for k, v in ev.items()}
        out, state["map"] = downsample_events(dev, ih, iw, oh, ow, change_map=state["map"])
        return {k: v.cpu().numpy() for k, v in out.items()}
    return run


def downsample_stream(reader, sink, input_height, input_width, output_height, output_width, downsampler, chunk=CHUNK):
    n = len(reader)
    full = n // chunk
    for i in range(full):
        ev = reader.read(i * chunk, (i + 1) * chunk)
        ev["p"] = 2 * ev["p"].astype("int8") - 1                 # :155
        sink.add(downsampler(ev, input_height, input_width, output_height, output_width))
    ev = reader.read(full * chunk, n)                            # :160-163: the tail keeps p in {0, 1}
    if len(ev["t"]):
        sink.add(downsampler(ev, input_height, input_width, output_height, output_width))
    return sink.finish()


def main(argv=None, downsampler=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--input_path", type=Path, required=True)
    p.add_argument("--output_path", type=Path, required=True)
    p.add_argument("--input_height", type=int, default=480)
    p.add_argument("--input_width", type=int, default=640)
    p.add_argument("--output_height", type=int, default=240)
    p.add_argument("--output_width", type=int, default=320)
    a = p.parse_args(argv)
    reader = H5Events(a.input_path) if a.input_path.suffix == ".h5" else NpzEvents(a.input_path)
    counts = downsample_stream(reader, EventSink(a.output_path), a.input_height, a.input_width, a.output_height,
                               a.output_width, downsampler or device_downsampler())
    print(f"{len(reader)} events -> {counts['t']} at {a.output_width}x{a.output_height}: {a.output_path}")
    return counts


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    main()
