#!/usr/bin/env python
"""bench.py -- events/sec and per-window latency of the DAGR event-graph hot path on MI355X.

One "step" = one pass of the hot path (format_data'd events -> graph build -> SplineConv stack + voxel pooling ->
decoded detection-head outputs [B,175,5+C] -> device-side confidence mask + NMS) over one batch of B synthetic 50 ms
event windows already resident in HBM.  `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches
it under torch.distributed.run (one rank per GPU) -- run by hand without that environment, `--gpus N` starts the N
ranks itself and refuses to run on fewer devices; windows are independent so ranks share nothing on the data path
(weak scaling) and RCCL is used once, to gather the run's detections (variable length).  Prints ONE JSON line on rank 0:

  value / ms_per_step   BASELINE config 2 on the synthetic 640x480 stream (dagr-s + resnet50 image branch), K steps
  events_only           the same K steps on the events-only model (config 1 shape): the hand-written path alone
  latency_ms            per-window latency (HIP events, one window batch at a time through the captured window graph, 20
                        warm-up + 100 timed windows): median / p95 for B in {1, 8}, N in {25k..400k} events per window,
                        S-uniform and S-edges
  async_update          f3: microseconds per reset=False update (1 / 10 / 100 events onto a 25 k-event window) vs re-evaluation
  roofline / stages     dominant kernel of the event path and per-stage timings (HIP events on the kernels' stream)
  cpu_baseline          the CPU oracle (a port of the reference's op sequence) on the host cores: median of 3 steps of a
                        bounded sample of the step's windows, for the headline model and (events_only.cpu_baseline) the
                        events-only one
  image_branch          the dense ResNet-50 branch's time, GFLOP and rate against the fp32 matrix peak (library code)
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: 6.29 TB/s measured float4 copy -- what `floor_us` is priced at


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--events-per-window", type=int, default=100000)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--stream", choices=["uniform", "edges"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the per-window latency sweep")
    ap.add_argument("--no-rccl", action="store_true", help="world 1: no one-rank process group (plain barriers, local gather)")
    ap.add_argument("--no-events-only-leg", action="store_true", help="skip the events-only sub-measurement")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the other-stream and the H2D-inclusive legs")
    ap.add_argument("--repeats", type=int, default=1,
                    help="builder's collections: time the headline region this many times more AFTER the line's own (the "
                         "line's `value` stays the first, exactly --steps steps); the spread goes into `repeat_ms_per_step`")
    ap.add_argument("--events-only", action="store_true",
                    help="make the events-only model (BASELINE config 1 shape) the headline instead of config 2")
    ap.add_argument("--img-net", default="resnet50")
    ap.add_argument("--engines", type=int, default=3,
                    help="independent engine instances (own buffers + stream) that consecutive window batches "
                         "rotate through; windows share no state, so batch i's latency-bound tail overlaps the level 0 "
                         "of batches i+1, i+2")
    ap.add_argument("--cpu-steps", type=int, default=5, help="steps the CPU baseline legs time after one untimed warm-up "
                                                             "step (median is reported; x --cpu-batch windows each)")
    ap.add_argument("--cpu-batch", type=int, default=2,
                    help="windows per CPU-baseline step (a bounded sample of the GPU step's --batch windows of the same "
                         "size; the oracle's cost is linear in the number of windows)")
    ap.add_argument("--latency-windows", type=int, default=100)
    ap.add_argument("--latency-warmup", type=int, default=20)
    ap.add_argument("--latency-n", type=str, default="25000,50000,100000,200000,400000")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="no device: the launch logic, rendezvous, barrier-bracketed timed region, detection gather and "
                         "per-rank report over gloo on CPU with a stand-in rig (value is null)")
    return ap.parse_args()


def source_stamp():
    """sha256 over the kernel sources + build flags: PMC traffic measured on another build is refused as stale."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "dagr_amd", "csrc", "*"))) + [os.path.join(ROOT, "Makefile")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher's environment: start N ranks of this same command line under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and return their exit code."""
    if not a.dry_run_gloo:
        n_dev = torch.cuda.device_count()
        if n_dev < a.gpus:
            raise SystemExit(f"bench.py: {a.gpus} ranks requested, {n_dev} device(s) visible -- refusing to run fewer "
                             f"ranks than asked for")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def make_model(W, H, B, use_image=False, img_net="resnet50", name="dagr-s", **over):
    from dagr_amd.utils.args import model_args   # config/dagr-s-dsec.yaml values (pinned to the reference's FLAGS())
    from dagr_amd.model.networks.dagr import DAGR
    from dagr_amd.utils.testing_weights import randomize_
    torch.manual_seed(0)
    args = model_args(name, batch_size=B, use_image=use_image, img_net=img_net, **over)   # (`over`: probes of other YAML keys)
    model = randomize_(DAGR(args, height=H, width=W)).eval()
    return args, model


def algorithmic_bytes(N, E, r, levels, use_image=False):
    """SURVEY.md 8(d) per-step algorithmic bytes (int32 indices, fp32 features, no LUT)."""
    def conv(cin, cout, nn, ee):
        return 4 * cin * ee + 8 * ee + 4 * (nn + 1) + 4 * cin * nn + 4 * cout * nn + 104 * cin * cout

    def sample(cf, nn):  # 4 taps read + write
        return 16 * cf * nn + 4 * cf * nn
    fch = [16, 64, 64, 64, 64] if use_image else [0] * 5
    c0 = 3 + fch[0]
    out = {}
    out["graph"] = 16 * N + 4 * (2 * r + 1) ** 2 * N + 12 * E + 4 * (N + 1)
    out["l0_input"] = sample(fch[0], N) + 4 * 3 * N
    out["l0_conv1"] = conv(c0, 16, N, E)
    out["l0_conv2"] = conv(16, 16, N, E) + 4 * c0 * N
    out["l0_sample1"] = sample(fch[1], N)
    n1, e1 = levels[0]
    cp = 16 + fch[1]
    out["pool1"] = 4 * (cp + 5) * N + 8 * E + 4 * (cp + 4) * n1 + 12 * e1      # 8(d): x, pos, batch in + edges in + level 1 out
    tail = 0
    for k, (nn, ee) in enumerate(levels):
        cin = (16 if k == 0 else 64) + fch[k + 1] + 2
        tail += conv(cin, 64, nn, ee) + conv(64, 64, nn, ee) + 4 * cin * nn
        if k < 3:
            tail += sample(fch[k + 2], nn)
            nc, ec = levels[k + 1]
            cpk = 64 + fch[k + 2]
            tail += 4 * (cpk + 5) * nn + 8 * ee + 4 * (cpk + 4) * nc + 12 * ec
    out["tail"] = tail
    return out


def compulsory_bytes(N, K, use_image):
    """Bytes a level-0 conv launch cannot avoid moving once (what `floor_us` prices): its input rows, the fixed-stride
    neighbour lists as this library stores them (K x (int32 src + int16 code) + int32 degree), its output rows; the
    per-edge gathers the 8(d) formula charges are served by L2 because level 0 is laid out in pixel-slot order."""
    c0 = 19 if use_image else 3
    nbr = K * 6 + 4
    return {"l0_conv1": N * (4 * c0 + nbr + 64), "l0_conv2": N * (64 + 4 * c0 + nbr + 64)}


def _sync():
    if torch.cuda.is_available():   # (the CPU/gloo dry-run of the control flow in tests/ has no device to wait for)
        torch.cuda.synchronize()


def time_gpu(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(iters):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / iters  # ms


def cpu_baseline(model_cpu, model_sd, W, H, B, n_events, n_steps, stream, use_image, img_net):
    """The oracle (op-for-op CPU restatement, `port`) on a bounded sample of THE SAME step the GPU line times: B windows
    of the same synthetic stream per step.  Graph build: the C restatement of the reference kernels, one thread per
    sample (windows share no state); conv / pool (and, with the image branch, the same torch ResNet / CNN-head modules)
    on all host threads."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import graph as og
    from oracle import model as om
    from dagr_amd.utils import synthetic as syn
    gen = syn.uniform_window if stream == "uniform" else syn.edges_window
    a = om.default_args(batch_size=B, use_image=use_image, img_net=img_net)
    nc = om.NetConstants(a, H, W)
    r, dt = og.graph_params(a.radius, W, 1000000)
    times, t_graph = [], []
    with torch.no_grad(), ThreadPoolExecutor(max_workers=min(B, os.cpu_count() or 1)) as pool:
        for w in range(-1, n_steps):          # step -1: one untimed warm-up step (allocator, thread pools, cold caches)
            x, y, t, p, b = syn.batch_windows(gen, n_events, B, W, H, seed=1234 + 10 * max(w, 0))
            t0 = time.perf_counter()
            image_feat = cnn_out = None
            if use_image:
                img = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8,
                                    generator=torch.Generator().manual_seed(w)).float() / 255     # format_data: uint8 / 255
                feats, outs = model_cpu.backbone.net(img)
                resized = [torch.nn.functional.interpolate(f, o) for f, o in zip(outs[-a.num_scales:], nc.output_sizes)]
                cnn_out = model_cpu.head.cnn_head(resized)
                image_feat = feats
            tg = time.perf_counter()
            dpos = og.denormalize_pos(syn.format_data_np(x, y, t, W, H), W, H, 1000000)
            lo = [int(np.searchsorted(b, s)) for s in range(B + 1)]

            def one(s):
                sl = slice(lo[s], lo[s + 1])
                return og.build_window_graph(dpos[sl, 0], dpos[sl, 1], dpos[sl, 2], np.zeros(lo[s + 1] - lo[s], np.int32),
                                             W, H, 1, r, dt, K=a.max_neighbors, Q=128) + lo[s]
            ei = np.concatenate(list(pool.map(one, range(B))), axis=1)
            tg = time.perf_counter() - tg
            om.forward_events(model_sd, a, H, W, x, y, t, p, b, B, image_feat=image_feat, cnn_out=cnn_out,
                              edge_index=torch.from_numpy(ei))
            if w >= 0:
                t_graph.append(tg)
                times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    what = f"dagr-s + {img_net} image branch" if use_image else "events-only dagr-s"
    return dict(value=B * n_events / med, unit="events/s", cores=torch.get_num_threads(), kind="port",
                windows_per_step=B, ms_per_window=round(1e3 * med / B, 1),   # (the GPU line's step holds gpu_windows_per_step)
                step_ms_median=round(1e3 * med, 1), step_ms_all=[round(1e3 * t, 1) for t in times],
                graph_ms_median=round(1e3 * float(np.median(t_graph)), 1), graph_threads=min(B, os.cpu_count() or 1),
                windows_timed=B * n_steps,
                sample=f"median of {n_steps} steps (after one untimed warm-up step) of B={B} windows x {n_events} events = "
                       f"{B * n_steps} windows, {W}x{H}, {what} (the GPU line's "
                       f"windows and model, {B} windows per step instead of the GPU step's batch); oracle/model.py "
                       f"(torch-CPU fp32 on {torch.get_num_threads()} threads + C graph builder, one thread per sample), "
                       f"{sum(times):.1f} s of CPU work")


class DryRunRig:
    """--dry-run-gloo stand-in for the model (CPU, no library): deterministic per-(rank, step) detections -- image b of
    step i on rank r keeps (r + i + b) % 4 rows -- so that launch, rendezvous, timed region, gather and report run
    exactly as on GPUs."""
    B, A = 3, 8

    class _Engine:
        def check_status(self):
            pass

    def __init__(self, rank):
        self.rank, self.dev = rank, torch.device("cpu")
        self.engines = [self._Engine()]
        self.streams = []

    def step(self, i, slots):
        det = torch.zeros((self.B, self.A, 6))
        n = torch.tensor([(self.rank + i + b) % 4 for b in range(self.B)], dtype=torch.int32)
        for b in range(self.B):
            for k in range(int(n[b])):
                det[b, k] = torch.tensor([1.0 * k, 2.0, 3.0 + k, 4.0, 0.5, float(self.rank)])
        return det, n

    def drain(self):
        pass


class Rig:
    """One model on the device + its engines / streams + resident synthetic slots."""

    def __init__(self, W, H, B, use_image, img_net, n_eng, dev, low_latency=False, model_name="dagr-s", **over):
        from dagr_amd.engine import WindowEngine
        self.W, self.H, self.B, self.use_image, self.dev = W, H, B, use_image, dev
        self.args, model = make_model(W, H, B, use_image=use_image, img_net=img_net, name=model_name, **over)
        self.sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
        self.model = model.to(dev)
        self.model.cache_luts(width=W, height=H, radius=self.args.radius)
        eng = self.model.engine()
        self.engines = [eng] + [WindowEngine(self.model) for _ in range(n_eng - 1)]
        for e in self.engines:       # throughput rigs keep the GPU full from several streams; latency rigs run one window
            e.set_low_latency(low_latency)
        self.streams = [torch.cuda.Stream(dev) for _ in range(n_eng)] if n_eng > 1 else [torch.cuda.current_stream(dev)]
        self.num_classes = self.model.backbone.num_classes

    def make_slots(self, gen, npw, n_slots, seed):
        from dagr_amd.utils import synthetic as syn
        slots = []
        for s in range(n_slots):
            x, y, t, p, b = syn.batch_windows(gen, npw, self.B, self.W, self.H, seed=seed + 10 * s)
            pos = torch.from_numpy(syn.format_data_np(x, y, t, self.W, self.H)).to(self.dev)
            feat = torch.from_numpy(p.astype(np.float32)).view(-1, 1).to(self.dev)
            batch = torch.from_numpy(b).to(self.dev)
            image = None
            if self.use_image:  # format_data'd frames (uint8/255, utils/buffers.py:37-38), resident like the events
                image = (torch.randint(0, 256, (self.B, 3, self.H, self.W), dtype=torch.uint8,
                                       generator=torch.Generator().manual_seed(77 + s + seed - 1234)).float() / 255).to(self.dev)
            slots.append((pos, feat, batch, image))
        return slots

    def step(self, i, slots):
        """forward + device-side post-processing of window batch i on engine i % n (engine.forward_detections: every image's
        confidence mask + NMS inside the heads' last launch); returns (det[B,A,6], n_keep[B])."""
        pos, feat, batch, image = slots[i % len(slots)]
        k = i % len(self.engines)
        e, st = self.engines[k], self.streams[k]
        with torch.cuda.stream(st):
            # (post-processed at once, on the engine's stream, before its next window: no copy of the output buffer)
            return e.forward_detections(pos, feat, batch, image=image)

    def drain(self):
        cur = torch.cuda.current_stream(self.dev)
        for st in self.streams:
            cur.wait_stream(st)


def h2d_run(rig, gen, npw, steps, warmup, seed):
    """The step as ``utils/testing.py:29`` pays it: every window batch starts in HOST memory in the loader's dtypes (pos
    int16[N,2], t int32[N], polarity int8[N], sample index int64[N], frame uint8[B,3,H,W]; pinned), is copied with
    ``non_blocking`` copies on a copy stream into one of two device staging sets, formatted on the device
    (``format_data``: dagr_format_events, frame / 255) and run through forward + post-processing.  The copy of batch i + 1
    overlaps the compute of batch i.  Returns (elapsed seconds of `steps` steps, bytes copied per step)."""
    from dagr_amd import _lib
    from dagr_amd.utils import synthetic as syn
    dev, B, W, H = rig.dev, rig.B, rig.W, rig.H
    host = []
    for s in range(4):
        x, y, t, p, b = syn.batch_windows(gen, npw, B, W, H, seed=seed + 10 * s)
        h = dict(xy=torch.from_numpy(np.stack([x, y], -1).astype(np.int16)).pin_memory(),
                 t=torch.from_numpy(t.astype(np.int32)).pin_memory(), p=torch.from_numpy(p.astype(np.int8)).pin_memory(),
                 b=torch.from_numpy(b.astype(np.int64)).pin_memory())
        if rig.use_image:
            h["img"] = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8,
                                     generator=torch.Generator().manual_seed(77 + s)).pin_memory()
        host.append(h)
    nbytes = sum(v.numel() * v.element_size() for v in host[0].values())
    n_eng = len(rig.engines)
    S = n_eng + 1                                # staging sets: one per engine in flight + the one being filled
    stage = [{k: torch.empty_like(v, device=dev) for k, v in host[0].items()} for _ in range(S)]
    copy_stream = torch.cuda.Stream(dev)
    copied = [torch.cuda.Event() for _ in range(S)]
    consumed = [torch.cuda.Event() for _ in range(S)]
    L = _lib.lib()

    def issue_copy(i):
        st = stage[i % S]
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[i % S])          # the staging set's previous batch has been formatted
            for k, v in host[i % len(host)].items():
                st[k].copy_(v, non_blocking=True)
            copied[i % S].record(copy_stream)

    def step(i):
        st = stage[i % S]
        k = i % n_eng
        eng, stream = rig.engines[k], rig.streams[k]
        with torch.cuda.stream(stream):
            stream.wait_event(copied[i % S])
            N = st["t"].shape[0]
            pos = torch.empty((N, 3), dtype=torch.float32, device=dev)
            feat = torch.empty((N, 1), dtype=torch.float32, device=dev)
            _lib.check(L.dagr_format_events(_lib.ptr(st["xy"]), _lib.ptr(st["t"]), _lib.ptr(st["p"]), N, W, H, 1000000,
                                            _lib.ptr(pos), _lib.ptr(feat), _lib.cur_stream(dev)), "format_events")
            image = st["img"].float() / 255.0 if rig.use_image else None
            batch = st["b"].clone()
            consumed[i % S].record(stream)
            return eng.forward_detections(pos, feat, batch, image=image)
    cur = torch.cuda.current_stream(dev)
    for e in consumed:
        e.record(cur)
    torch.cuda.synchronize()
    issue_copy(0)
    for i in range(warmup):
        issue_copy(i + 1)
        step(i)
    rig.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        issue_copy(i + 1)
        step(i)
    rig.drain()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, nbytes


def detections_rows(results, rank, B):
    """Variable-length detection rows (window id, x1, y1, x2, y2, score, label) of a run, cut on the device."""
    det = torch.stack([d for d, _ in results])            # [K, B, A, 6]
    n = torch.stack([k for _, k in results])               # [K, B]
    K, _, A, _ = det.shape
    wid = (torch.arange(K, device=det.device).view(K, 1) * B + torch.arange(B, device=det.device).view(1, B)
           + rank * K * B).float()
    rows = torch.cat([wid.view(K, B, 1, 1).expand(K, B, A, 1), det], -1)
    mask = torch.arange(A, device=det.device).view(1, 1, A) < n.view(K, B, 1)
    return rows[mask]


def timed_run(rig, slots, steps, warmup, dist, world, rank):
    """W untimed warm-up steps, then exactly `steps` steps between barrier + synchronize brackets; MAX over ranks."""
    from dagr_amd import parallel
    dev = rig.dev
    warm = [rig.step(i, slots) for i in range(warmup)]
    if warm:   # the detection cut + gather once outside the timed region (allocator / lazy-init costs of its torch ops)
        rows = detections_rows(warm, rank, rig.B)
        if dist is not None:
            parallel.gather_detections(rows)
    del warm
    _sync()
    for e in rig.engines:
        e.check_status()
    _sync()
    if dist is not None:
        dist.barrier()
    _sync()
    t0 = time.perf_counter()
    results = [rig.step(i, slots) for i in range(steps)]
    rig.drain()
    _sync()
    t_compute = time.perf_counter() - t0
    rows = detections_rows(results, rank, rig.B)           # the run's detections of this rank
    allrows = parallel.gather_detections(rows) if dist is not None else rows   # the job's only collective (RCCL)
    _sync()
    t_gather = time.perf_counter() - t0 - t_compute
    if dist is not None:
        dist.barrier()
    _sync()
    elapsed = time.perf_counter() - t0
    per_rank, ranks_seen = None, 1
    if dist is not None:
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones)                               # every rank of the job took part in the collectives
        ranks_seen = int(ones.item())
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([t_compute, t_gather], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [[float(v) for v in t.tolist()] for t in every]
    rig.engines[0].check_status()
    return dict(elapsed=elapsed, t_compute=t_compute, t_gather=t_gather, per_rank=per_rank, ranks_seen=ranks_seen,
                n_detections=int(allrows.shape[0]))


def stage_timings(rig, slots, n_events_step):
    """Per-stage / per-kernel timing of one engine on the stream the kernels run on (HIP events) + roofline inputs."""
    eng = rig.engines[0]
    use_image = rig.use_image
    pos, feat, batch, image = slots[0]
    eng.forward_raw(pos, feat, batch, image=image)
    ne, _ = eng.graph.status()
    levels = [tuple(int(v) for v in lvl.counts.tolist()) for lvl in eng.levels]
    r = eng.graph.params["radius"]
    ab = algorithmic_bytes(n_events_step, ne, r, levels, use_image)
    iters = 20
    stages = {}
    if use_image:
        stages["image_branch"] = time_gpu(lambda: eng.stage_image(image), 5, warm=1)
    stages["graph"] = time_gpu(lambda: eng.stage_graph(pos, batch), iters)
    # the neighbour search alone (k_search_rows + the sweep of its deferral list), on the pixel index just built
    stages["graph_search"] = time_gpu(lambda: eng.graph.search_again(eng._nbr), iters)
    eng.stage_l0_input(feat)
    stages["l0_input"] = time_gpu(lambda: eng.stage_l0_input(feat), iters)
    stages["l0_conv1"] = time_gpu(eng.stage_l0_conv1, iters)
    stages["l0_conv2"] = time_gpu(lambda: eng.stage_l0_conv2(sample=False), iters)
    if use_image:   # the sampling kernel itself (sampling_skip(image_feat[1]) into hp0's columns 16..), not a difference
        stages["l0_sample1"] = time_gpu(eng.stage_l0_sample1, iters)
    stages["pool1"] = time_gpu(eng.stage_pool1, iters)
    # pool1's accumulation kernel alone (the one HBM-proportional launch of the stage; scan + emit work on the voxel table)
    stages["pool1_accumulate"] = time_gpu(eng.pool1_accumulate_again, iters)
    eng.stage_pool1()            # re-arms the accumulators the repeated launches filled
    stages["tail"] = time_gpu(eng.stage_tail, iters)
    stages["head"] = time_gpu(eng.stage_head, iters)
    if not use_image:
        # what a latency-mode engine runs after pool1: tail + both heads + decode as ONE replayed HIP graph (head scale 1
        # beside pool4 / layer5 / head scale 2); `tail` and `head` above are the same kernels launched one by one
        eng.set_low_latency(True)
        stages["tail_head_graph"] = time_gpu(eng._replay_tail, iters, warm=4)
        eng.set_low_latency(False)
    kernels = {k: dict(ms=round(v, 4), alg_MB=round(ab[k] / 1e6, 2) if k in ab else None,
                       alg_GBs=round(ab[k] / 1e9 / (v / 1e3), 1) if k in ab else None)
               for k, v in stages.items()}
    if "l0_sample1" in kernels:
        # 8(d) charges four taps per (node, channel); on this layout (nodes in pixel order) four of every five of those reads
        # are cache hits, so the formula's bytes over the time is not an HBM rate (it printed > 8 TB/s).  Charged instead:
        # what the kernel has to move once -- the feature map in, one row of Cf floats per node out; the formula's figure
        # stays beside it as `formula_MB`.
        fmap = eng._img_feats[1]
        cf = 64
        comp_b = fmap.numel() * fmap.element_size() + 4 * cf * n_events_step + 16 * n_events_step
        kernels["l0_sample1"] = dict(ms=kernels["l0_sample1"]["ms"], alg_MB=round(comp_b / 1e6, 2),
                                     alg_GBs=round(comp_b / 1e9 / (stages["l0_sample1"] / 1e3), 1),
                                     formula_MB=round(ab["l0_sample1"] / 1e6, 2), bytes="compulsory (map once + rows out)")
    # the dominant kernel of the event path: the longest single-kernel stage among ALL of them (the other stages are
    # sequences of short launches, reported as stages)
    N_ = n_events_step
    ab["graph_search"] = 4 * (2 * r + 1) ** 2 * N_ + 12 * ne + 4 * (N_ + 1)     # 8(d): FIFO probes + timestamps + edges
    comp = compulsory_bytes(N_, eng.graph.K, use_image)
    # what this design's search cannot avoid moving once: the per-pixel offsets, {id, t} + x|y|b per event, its lists
    comp["graph_search"] = 4 * rig.W * rig.H * rig.B + 12 * N_ + 6 * ne + 4 * N_
    kernels["graph_search"] = dict(ms=round(stages["graph_search"], 4), alg_MB=round(ab["graph_search"] / 1e6, 2),
                                   alg_GBs=round(ab["graph_search"] / 1e9 / (stages["graph_search"] / 1e3), 1))
    # pool1's accumulation kernel streams every level-0 row once: input rows + positions + ids + degrees + neighbour
    # codes + slot words in, the (small) level-1 arrays out
    cp = 16 + (64 if use_image else 0)
    comp["pool1"] = N_ * (4 * cp + 12 + 8 + 4 + 2 * eng.graph.K + 4) + 4 * (cp + 4) * levels[0][0] + 12 * levels[0][1]
    ab["pool1_accumulate"] = ab["pool1"]
    comp["pool1_accumulate"] = comp.pop("pool1")
    cands = ("graph_search", "l0_conv1", "l0_conv2", "pool1_accumulate")
    dom = max(cands, key=lambda k: stages[k])
    # (the row search kernel as the product library launches it: graph_build.hip:launch_search)
    rows_name = "k_search_rows<320, 4, 6>"
    names = dict(eng.l0_kernel_names(), graph_search=rows_name,
                 pool1_accumulate=f"k_pool_l0_slots<0, {4 if cp % 4 == 0 else 1}>")
    kname = names[dom]
    # fp32 work of a level-0 conv launch (conv_l0_tiles.hip): phase 1 updates ALL taps of the TX x TY window per edge
    # (executed) although an edge weights only 4 of them (useful); phase 2 contracts [(NT + 1) * cin + cskip] x 16 per node
    nt = eng.ntaps0
    c0 = 3 + (16 if use_image else 0)

    def conv_flops(cin, cskip):
        phase2 = 2.0 * N_ * ((nt + 1) * cin + cskip) * 16
        return 2.0 * ne * nt * cin + phase2, 2.0 * ne * 4 * cin + phase2
    flops = {"l0_conv1": conv_flops(c0, 0), "l0_conv2": conv_flops(16, c0)}
    FP32_PEAK_TF = 157.3      # MI355X_MICROARCH.md: fp32 vector / matrix peak (the f32 MFMA runs at the VALU FMA rate)
    # HBM bytes per launch: PMC counters cannot be read from inside this process; the number is the one the
    # rocprofv3 --pmc passes of this same command produced (tools/pmc.sh -> profiles/r*_traffic.json, committed) --
    # accepted only if that file was measured on this build of the kernels (source stamp), else null
    stamp = source_stamp()
    tj_cfg, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        tj = json.load(open(path))
        if n_events_step != 800000 or (rig.W, rig.H) != (640, 480):
            break
        if tj.get("source_stamp") != stamp:
            src = f"stale: {os.path.basename(path)} was measured on another build of the kernels"
            continue
        tj_cfg, src = tj["configs"]["use_image" if use_image else "events_only"], "profiles/" + os.path.basename(path)
        break

    def describe(k):
        """One kernel against the roofline it is on: the level-0 convs are bound by their fp32 FMAs (the matrix pipe runs
        f32 at the vector rate: `mfma` against the 157.3 TF fp32 peak), search and pooling by HBM bytes."""
        ms = stages[k]
        traffic = tj_cfg[names[k]].get("traffic_bytes") if tj_cfg is not None and names[k] in tj_cfg else None
        frac_bytes = ab[k] / 1e9 / (ms / 1e3) / HBM_PEAK_GBS
        out = dict(kernel=names[k], stage=k, launch_ms=round(ms, 4), alg_bytes_per_launch=int(ab[k]),
                   frac_alg_bytes=round(frac_bytes, 4), traffic=traffic,
                   frac_traffic=(round(traffic / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4) if traffic else None),
                   compulsory_bytes=int(comp[k]), floor_us=round(comp[k] / (HBM_ACHIEVABLE_GBS * 1e3), 1),
                   x_over_floor=round(ms * 1e3 / (comp[k] / (HBM_ACHIEVABLE_GBS * 1e3)), 1))
        if k in flops:
            ex, useful = flops[k]
            tf = ex / 1e12 / (ms / 1e3)
            out.update(bound="mfma", achieved=round(tf, 1), peak=FP32_PEAK_TF, unit="TFLOP/s", frac=round(tf / FP32_PEAK_TF, 4),
                       executed_gflop=round(ex / 1e9, 2), useful_gflop=round(useful / 1e9, 2),
                       frac_useful=round(useful / 1e12 / (ms / 1e3) / FP32_PEAK_TF, 4))
        else:
            out.update(bound="hbm", achieved=round(ab[k] / 1e9 / (ms / 1e3), 1), peak=HBM_PEAK_GBS, unit="GB/s",
                       frac=round(frac_bytes, 4))
        return out
    roofline = describe(dom)
    roofline.update(traffic_source=src, source_stamp=stamp,
                    candidates={k: {f: v for f, v in describe(k).items()
                                    if f in ("kernel", "launch_ms", "bound", "achieved", "unit", "frac", "frac_alg_bytes",
                                             "frac_traffic", "x_over_floor", "frac_useful")} for k in cands})
    image_branch = None
    if use_image:
        # the dense image branch is library code (MIOpen / hipBLASLt fp32): its share and its rate against the fp32 matrix peak
        from torch.utils.flop_counter import FlopCounterMode
        with FlopCounterMode(display=False) as fc:
            feats_, outs_ = rig.model.backbone.net(image)
            rig.model.head.cnn_head([torch.nn.functional.interpolate(f, o) for f, o in
                                     zip(outs_[-eng.num_scales:], eng.out_sizes)])
        gflop = fc.get_total_flops() / 1e9
        image_branch = dict(ms=round(stages["image_branch"], 4), gflop=round(gflop, 1),
                            tflops=round(gflop / stages["image_branch"], 1),
                            frac_of_157=round(gflop / stages["image_branch"] / 157.3, 3),
                            note="ResNet-50 HookModule + CNNHead on PyTorch-ROCm (MIOpen / hipBLASLt fp32), FLOPs counted "
                                 "on the plain modules; not hand-written code")
    total = sum(v for k, v in stages.items() if k not in ("tail_head_graph", "graph_search", "pool1_accumulate"))
    return dict(roofline=roofline, stages=kernels, edges_per_step=int(ne), levels=levels, radius=r,
                batch_latency_ms=round(total, 4), image_branch=image_branch)


def latency_sweep(W, H, use_image, img_net, dev, Ns, n_warm, n_timed):
    """Per-window latency: one window batch at a time through ONE engine; the device is idle when a window starts
    (synchronize), HIP events bracket forward + post-processing on the engine's stream."""
    from dagr_amd.utils import synthetic as syn
    out = {}
    for B in (1, 8):
        rig = Rig(W, H, B, use_image, img_net, 1, dev, low_latency=True)
        eng = rig.engines[0]
        for sname, gen in (("uniform", syn.uniform_window), ("edges", syn.edges_window)):
            for N in Ns:
                slots = rig.make_slots(gen, N, 2, seed=4234)

                def window(i):
                    pos, feat, batch, image = slots[i % 2]
                    # forward + post-processing as DAGR.forward runs them in latency mode: ONE captured graph from the
                    # staged events to det[B, A, 6] + n_keep[B] (engine.forward_detections; thresholds = the model's)
                    return eng.forward_detections(pos, feat, batch, image=image)
                for i in range(n_warm):
                    window(i)
                torch.cuda.synchronize()
                evs = []
                for i in range(n_timed):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    window(i)
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                eng.check_status()
                ms = np.array([a.elapsed_time(b) for a, b in evs])
                out.setdefault(sname, {}).setdefault(f"B{B}", {})[str(N)] = dict(
                    p50=round(float(np.median(ms)), 4), p95=round(float(np.percentile(ms, 95)), 4),
                    events_per_s=round(B * N / (float(np.median(ms)) / 1e3), 1))
                del slots
        del rig, eng
        torch.cuda.empty_cache()
    return out


def dry_run(a, world, rank, rank_env, t_start):
    """--dry-run-gloo: every step of a multi-rank run except the model, on CPU."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    envs = [None] * world
    dist.all_gather_object(envs, dict(rank=rank, miopen_db=rank_env["miopen_db"], cores=rank_env["cores"],
                                      startup_s=round(time.perf_counter() - t_start, 3)))
    run = timed_run(DryRunRig(rank), None, a.steps, a.warmup, dist, world, rank)
    if rank == 0:
        print(json.dumps({"metric": "events_per_sec", "value": None, "unit": "events/s", "n_gpus": world, "world": world,
                          "ranks_seen": run["ranks_seen"], "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(1e3 * run["elapsed"] / a.steps, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "dry-run (gloo, no device)",
                          "config": {"workload": "control flow only: DryRunRig"},
                          "gather": {"detections": run["n_detections"]}, "rank_env": envs,
                          "per_rank": [dict(compute_ms=round(1e3 * c, 3), gather_ms=round(1e3 * g, 3))
                                       for c, g in run["per_rank"]]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def async_update_leg(W, H, dev, n_window=25000, sizes=(1, 10, 100), n_updates=100, n_warm=10):
    """f3: microseconds per asynchronous update -- a micro-batch of m events attaching to a resident 25 k-event window
    (B = 1, events-only, latency mode), p50 / p95 over `n_updates` consecutive updates, HIP events on the engine's stream,
    device idle at the start of each update; beside it the re-evaluation of the whole window (what reset=False cost before
    the incremental update existed)."""
    from dagr_amd.utils import synthetic as syn
    rig = Rig(W, H, 1, False, "resnet50", 1, dev, low_latency=True)
    eng = rig.engines[0]
    n_extra = max(sizes) * (n_updates + n_warm)
    x, y, t, p = syn.uniform_window(n_window + n_extra, W, H, seed=4234)
    pos = torch.from_numpy(syn.format_data_np(x, y, t, W, H)).to(dev)
    feat = torch.from_numpy(p.astype(np.float32)).view(-1, 1).to(dev)
    batch = torch.zeros(len(x), dtype=torch.int64, device=dev)

    def timed(fn, n):
        ms = []
        for i in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fn(i)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        return np.array(ms)

    out = {"window_events": n_window, "protocol": f"{n_warm} warm-up + {n_updates} timed updates per micro-batch size, one at "
                                                   "a time, HIP events around update + post-processing"}
    full = timed(lambda i: eng.forward_detections(pos[:n_window], feat[:n_window], batch[:n_window]), 30)[10:]
    out["reevaluate_window_us"] = dict(p50=round(1e3 * float(np.median(full)), 1), p95=round(1e3 * float(np.percentile(full, 95)), 1))
    for m in sizes:
        eng.forward_raw(pos[:n_window], feat[:n_window], batch[:n_window])

        def upd(i, m=m):
            lo = n_window + i * m
            eng.forward_detections(pos[lo:lo + m], feat[lo:lo + m], batch[lo:lo + m], append=True)
        us = 1e3 * timed(upd, n_warm + n_updates)[n_warm:]
        eng.check_status()
        out[f"update_{m}_events_us"] = dict(p50=round(float(np.median(us)), 1), p95=round(float(np.percentile(us, 95)), 1),
                                            events_per_s=round(m / (float(np.median(us)) * 1e-6), 1))
    del rig
    torch.cuda.empty_cache()
    return out


def _claim_stdout():
    """The run's ONE JSON line goes to the process's real stdout; everything else that writes to file descriptor 1 on the way
    (RCCL prints its version banner there when a communicator is made) is sent to stderr.  Returns the real stdout's fd."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def _emit(real_fd, text):
    sys.stdout.flush()
    os.write(real_fd, (text + "\n").encode())


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))            # N ranks of this command line, one per GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    t_start = time.perf_counter()
    from dagr_amd import parallel as _parallel
    rank_env = _parallel.rank_environment()         # per-rank MIOpen db + core slice, before anything touches the device
    if a.dry_run_gloo:
        return dry_run(a, world, rank, rank_env, t_start)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    real_out = _claim_stdout()
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device ({torch.cuda.device_count()} visible, local rank {local_rank})")
    torch.backends.cudnn.benchmark = True   # MIOpen picks its fastest fp32 conv kernels for the image branch
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rccl_note = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    elif not a.no_rccl and all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
        # one rank under a launcher (python -m torch.distributed.run --nproc-per-node 1): a one-rank RCCL group all the same,
        # so that the barriers and the detection gather of the timed region are the code -- and the library -- of the
        # N > 1 runs.  Without a launcher (python bench.py) there is no group: plain barriers, local gather.
        import torch.distributed as tdist
        try:
            tdist.init_process_group("nccl", device_id=dev)
            dist = tdist
            rccl_note = "one-rank nccl group (launcher rendezvous)"
        except Exception as exc:                      # the bench line must not depend on it
            rccl_note = f"no process group at world 1 ({type(exc).__name__}: {exc})"[:200]
    from dagr_amd.utils import synthetic as syn

    W, H, B, NPW = a.width, a.height, a.batch, a.events_per_window
    use_image = not a.events_only
    n_eng = max(1, a.engines)
    gen = syn.uniform_window if a.stream == "uniform" else syn.edges_window
    n_events_step = B * NPW
    ctx = torch.no_grad()
    ctx.__enter__()

    # ---- headline: the configuration BASELINE.json quotes the metric on
    rig = Rig(W, H, B, use_image, a.img_net, n_eng, dev)
    slots = rig.make_slots(gen, NPW, 4, seed=1234 + 1000 * rank)
    run = timed_run(rig, slots, a.steps, a.warmup, dist, world, rank)

    # ---- the hand-written path alone: same steps on the events-only model
    ev_leg = ev_rig = ev_slots = None
    if use_image and not a.no_events_only_leg:
        ev_rig = Rig(W, H, B, False, a.img_net, n_eng, dev)
        ev_slots = ev_rig.make_slots(gen, NPW, 4, seed=1234 + 1000 * rank)
        ev_leg = timed_run(ev_rig, ev_slots, a.steps, a.warmup, dist, world, rank)

    repeats = [timed_run(rig, slots, a.steps, a.warmup, None, 1, rank)["elapsed"] for _ in range(max(0, a.repeats - 1))] \
        if world == 1 else []

    # ---- the same step on the other synthetic stream (S-edges when the line is S-uniform and vice versa) and with the
    # window batches starting in host memory (what the reference's loop pays per batch, utils/testing.py:29)
    other = h2d = None
    if world == 1 and not a.no_side_legs:
        other_gen = syn.edges_window if a.stream == "uniform" else syn.uniform_window
        o_slots = rig.make_slots(other_gen, NPW, 4, seed=1234 + 1000 * rank)
        other = timed_run(rig, o_slots, a.steps, a.warmup, None, 1, rank)
        del o_slots
        h2d = h2d_run(rig, gen, NPW, a.steps, a.warmup, seed=1234 + 1000 * rank)

    result = None
    if rank == 0:
        ms_per_step = 1e3 * run["elapsed"] / a.steps
        value = world * n_events_step * a.steps / run["elapsed"]
        st = stage_timings(rig, slots, n_events_step)
        result = {
            "metric": "events_per_sec", "value": round(value, 1), "unit": "events/s", "n_gpus": world,
            "world": world, "ranks_seen": run["ranks_seen"], "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("dagr-s + --use_image --img_net " + a.img_net if use_image else "dagr-s events-only")
                                   + f", {W}x{H} synthetic S-{a.stream}, B={B} windows/step x {NPW} events (50 ms "
                                   f"each), r={st['radius']}, K=16, events->graph->GNN(+image fusion)->decoded head "
                                   "outputs->confidence mask + NMS (device)",
                       "image_branch": ("in-line" if use_image else None), "engines": n_eng,
                       "events_per_step_per_gpu": n_events_step, "edges_per_step": st["edges_per_step"],
                       "level_nodes_edges": st["levels"],
                       # one batch through one engine, stages run back to back (sum of the isolated stage timings
                       # below; the measured per-window latency is `latency_ms`)
                       "stage_sum_ms": st["batch_latency_ms"]},
            "gather": {"detections": run["n_detections"], "gather_ms_rank0": round(1e3 * run["t_gather"], 3),
                       "compute_ms_rank0": round(1e3 * run["t_compute"], 3),
                       "backend": ("nccl (RCCL)" if dist is not None else "none"), "note": rccl_note},
            "roofline": st["roofline"], "stages": st["stages"],
        }
        if repeats:
            result["repeat_ms_per_step"] = [round(ms_per_step, 4)] + [round(1e3 * r / a.steps, 4) for r in repeats]
        if other is not None:
            oname = "edges" if a.stream == "uniform" else "uniform"
            result["value_" + oname] = round(n_events_step * a.steps / other["elapsed"], 1)
            result["ms_per_step_" + oname] = round(1e3 * other["elapsed"] / a.steps, 4)
        if h2d is not None:
            # the line's engines, copies on a copy stream under the previous batches' compute; `value` above has its inputs resident
            result["value_h2d"] = round(n_events_step * a.steps / h2d[0], 1)
            result["h2d"] = {"ms_per_step": round(1e3 * h2d[0] / a.steps, 4), "MB_per_step": round(h2d[1] / 1e6, 2),
                             "engines": n_eng, "protocol": "pinned host batches in the loader's dtypes (int16 xy, int32 t, int8 "
                             "polarity, int64 sample index, uint8 frames) -> non_blocking copies on a copy stream into two "
                             "staging sets -> format_data on the device -> forward + post-processing; copy i+1 under compute i"}
        if st["image_branch"] is not None:
            ib = dict(st["image_branch"])
            ib["share_of_step"] = round(ib["ms"] / ms_per_step, 3)     # one engine's isolated stage time over the step time
            result["image_branch"] = ib
        if run["per_rank"] is not None:
            result["per_rank"] = [dict(events_per_s=round(n_events_step * a.steps / c, 1), gather_ms=round(1e3 * g, 3))
                                  for c, g in run["per_rank"]]
        if ev_leg is not None:
            est = stage_timings(ev_rig, ev_slots, n_events_step)
            result["events_only"] = {
                "value": round(world * n_events_step * a.steps / ev_leg["elapsed"], 1), "unit": "events/s",
                "ms_per_step": round(1e3 * ev_leg["elapsed"] / a.steps, 4), "steps": a.steps, "engines": n_eng,
                "workload": f"dagr-s events-only, {W}x{H} S-{a.stream}, B={B} x {NPW} events",
                "roofline": est["roofline"], "stages": est["stages"], "stage_sum_ms": est["batch_latency_ms"]}
    want_cpu = rank == 0 and world == 1 and not a.no_cpu_baseline
    sd_cpu = rig.sd_cpu
    ev_sd_cpu = ev_rig.sd_cpu if ev_rig is not None else None
    model_cpu = None
    if want_cpu and use_image:
        _, model_cpu = make_model(W, H, B, use_image=True, img_net=a.img_net)
        model_cpu.load_state_dict(sd_cpu)
    del slots, ev_slots, rig, ev_rig
    torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not a.no_latency:
        Ns = [int(v) for v in a.latency_n.split(",") if v]
        lat = {"protocol": f"one window batch at a time on one engine in latency mode (the caller's events staged by one "
                           f"launch, the whole window -- image branch, graph build, level 0, tail, heads, decode -- replayed "
                           f"as one captured HIP graph sized for the engine's event capacity, then the post-processing "
                           f"launch), device idle at window start, HIP events around forward + post-processing; "
                           f"{a.latency_warmup} warm-up + {a.latency_windows} timed windows"}
        lat["events_only"] = latency_sweep(W, H, False, a.img_net, dev, Ns, a.latency_warmup, a.latency_windows)
        if use_image:
            lat["image_" + a.img_net] = latency_sweep(W, H, True, a.img_net, dev, Ns, a.latency_warmup,
                                                      a.latency_windows)
        result["latency_ms"] = lat
        result["async_update"] = async_update_leg(W, H, dev)
    if want_cpu:
        Bc = max(1, min(B, a.cpu_batch))
        result["cpu_baseline"] = cpu_baseline(model_cpu, sd_cpu, W, H, Bc, NPW, a.cpu_steps, a.stream, use_image,
                                              a.img_net)
        # a BOUNDED SAMPLE of the step: fewer windows per step than the GPU line, hence fewer graph-builder
        # threads and smaller torch batches; per-window times of both legs side by side
        result["cpu_baseline"]["gpu_windows_per_step"] = B
        result["cpu_baseline"]["gpu_ms_per_window"] = round(result["ms_per_step"] / B, 4)
        if ev_sd_cpu is not None and "events_only" in result:
            # BASELINE config 1's shape (events-only dagr-s on the CPU) beside the events_only leg of the GPU line
            result["events_only"]["cpu_baseline"] = cpu_baseline(None, ev_sd_cpu, W, H, Bc, NPW, a.cpu_steps, a.stream,
                                                                 False, a.img_net)
    if rank == 0:
        _emit(real_out, json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
