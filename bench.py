#!/usr/bin/env python
"""bench.py -- events/sec of the DAGR event-graph hot path on MI355X.

One "step" = one pass of the hot path (format_data'd events -> graph build -> SplineConv stack +
voxel pooling -> decoded detection-head outputs [B,175,5+C]) over one batch of B synthetic 50 ms
event windows already resident in HBM.  `python bench.py --gpus N --steps K --warmup W`; for N > 1
the driver launches it under torch.distributed.run (one rank per GPU); windows are independent so
ranks share nothing on the data path (weak scaling) and RCCL is used once, to all-gather the
detections of the run.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


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
    ap.add_argument("--events-only", action="store_true",
                    help="BASELINE config 1 shape (no --use_image) instead of config 2 (ResNet-50 image branch)")
    ap.add_argument("--img-net", default="resnet50")
    ap.add_argument("--engines", type=int, default=3,
                    help="independent engine instances (own buffers + stream) that consecutive window batches "
                         "rotate through; windows share no state, so batch i's latency-bound tail overlaps the level 0 "
                         "of batches i+1, i+2 (measured events-only: 1 -> 458 M, 2 -> 557 M, 3 -> 596 M, 4 -> 544 M ev/s)")
    ap.add_argument("--pipeline-image", dest="pipeline_image", action="store_true",
                    help="run the image branch one step ahead on a shared side stream instead of in-line per engine "
                         "(measured slower: 8.9 vs 8.2 ms/step with 2 engines)")
    ap.add_argument("--cpu-windows", type=int, default=4)
    return ap.parse_args()


def make_model(W, H, B, use_image=False, img_net="resnet50"):
    from dagr_amd.utils.args import model_args   # config/dagr-s-dsec.yaml values (pinned to the reference's FLAGS())
    from dagr_amd.model.networks.dagr import DAGR
    from dagr_amd.utils.testing_weights import randomize_
    torch.manual_seed(0)
    args = model_args("dagr-s", batch_size=B, use_image=use_image, img_net=img_net)
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
    out["pool1"] = 4 * (cp + 5) * N + 8 * E + 4 * (cp + 4) * n1 + 12 * e1
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


def cpu_baseline(model_cpu, model_sd, W, H, n_events, n_windows, stream, use_image, img_net):
    """The oracle (op-for-op CPU restatement, `port`) on a bounded sample: B=1 windows of the same
    synthetic stream.  Graph build: single-threaded C; conv/pool (and, with the image branch, the same
    torch ResNet/CNN-head modules) on all host threads."""
    from oracle import model as om
    from dagr_amd.utils import synthetic as syn
    gen = syn.uniform_window if stream == "uniform" else syn.edges_window
    a = om.default_args(batch_size=1, use_image=use_image, img_net=img_net)
    nc = om.NetConstants(a, H, W)
    tot_ev, t0 = 0, time.perf_counter()
    with torch.no_grad():
        for w in range(n_windows):
            x, y, t, p = gen(n_events, W, H, seed=1234 + w)
            b = np.zeros(len(x), np.int64)
            image_feat = cnn_out = None
            if use_image:
                img = torch.rand((1, 3, H, W), generator=torch.Generator().manual_seed(w))
                feats, outs = model_cpu.backbone.net(img)
                resized = [torch.nn.functional.interpolate(f, o) for f, o in zip(outs[-a.num_scales:], nc.output_sizes)]
                cnn_out = model_cpu.head.cnn_head(resized)
                image_feat = feats
            om.forward_events(model_sd, a, H, W, x, y, t, p, b, 1, image_feat=image_feat, cnn_out=cnn_out)
            tot_ev += len(x)
    dt = time.perf_counter() - t0
    what = f"dagr-s + {img_net} image branch" if use_image else "events-only dagr-s"
    return dict(value=tot_ev / dt, unit="events/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n_windows} windows x {n_events} events, {W}x{H}, B=1, {what}, "
                       f"oracle/model.py (torch-CPU fp32 + C graph builder), {dt:.1f} s")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    torch.backends.cudnn.benchmark = True   # MIOpen picks its fastest fp32 conv kernels for the image branch
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from dagr_amd.utils import synthetic as syn

    W, H, B, NPW = a.width, a.height, a.batch, a.events_per_window
    use_image = not a.events_only
    args, model = make_model(W, H, B, use_image=use_image, img_net=a.img_net)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model_cpu = None
    if use_image and world == 1 and not a.no_cpu_baseline:
        import copy
        model_cpu = copy.deepcopy(model).eval()
    model = model.to(dev)
    model.cache_luts(width=W, height=H, radius=args.radius)
    eng = model.engine()
    from dagr_amd.engine import WindowEngine
    n_eng = max(1, a.engines)
    engines = [eng] + [WindowEngine(model) for _ in range(n_eng - 1)]
    # one stream per engine; with --pipeline-image the image branch runs one batch ahead on a shared stream
    streams = [torch.cuda.Stream(dev) for _ in range(n_eng)] if n_eng > 1 else [torch.cuda.current_stream(dev)]
    img_stream = torch.cuda.Stream(dev) if use_image else None

    # synthetic inputs, resident in HBM before the timed region (distinct per rank and per slot)
    gen = syn.uniform_window if a.stream == "uniform" else syn.edges_window
    slots = []
    for s in range(4):
        x, y, t, p, b = syn.batch_windows(gen, NPW, B, W, H, seed=1234 + 1000 * rank + 10 * s)
        pos = torch.from_numpy(syn.format_data_np(x, y, t, W, H)).to(dev)
        feat = torch.from_numpy(p.astype(np.float32)).view(-1, 1).to(dev)
        batch = torch.from_numpy(b).to(dev)
        image = None
        if use_image:  # format_data'd frames (uint8/255, utils/buffers.py:37-38), resident like the events
            image = torch.rand((B, 3, H, W), generator=torch.Generator().manual_seed(77 + s + 10 * rank)).to(dev)
        slots.append((pos, feat, batch, image))
    n_events_step = B * NPW

    pending = {}   # step index -> image-branch handle started one step ahead on the side stream

    def step(i):
        s = i % len(slots)
        pos, feat, batch, image = slots[s]
        k = i % n_eng
        e, st = engines[k], streams[k]
        if n_eng > 1:
            if use_image and a.pipeline_image:
                h = pending.pop(i, None) or e.image_async(image, stream=img_stream)
                nxt = engines[(i + 1) % n_eng]
                pending[i + 1] = nxt.image_async(slots[(i + 1) % len(slots)][3], stream=img_stream)
                with torch.cuda.stream(st):
                    return e.forward_raw(pos, feat, batch, image_handle=h)
            with torch.cuda.stream(st):
                return e.forward_raw(pos, feat, batch, image=image)
        if use_image and a.pipeline_image:
            h = pending.pop(i, None) or eng.image_async(image)
            pending[i + 1] = eng.image_async(slots[(i + 1) % len(slots)][3])   # next batch's frames
            return eng.forward_raw(pos, feat, batch, image_handle=h)
        return eng.forward_raw(pos, feat, batch, image=image)

    ctx = torch.no_grad()
    ctx.__enter__()
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    for e in engines:
        e.check_status()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for i in range(a.steps):
        outs.append(step(i))
    for st in streams + ([img_stream] if img_stream is not None else []):
        torch.cuda.current_stream(dev).wait_stream(st)
    if dist is not None:  # the only collective of the job: gather the run's detections (RCCL)
        mine = outs[-1].contiguous()
        gathered = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=dev)
        dist.all_gather_into_tensor(gathered, mine)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    eng.check_status()

    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / a.steps
        value = world * n_events_step * a.steps / elapsed
        # ---- per-stage / per-kernel timing on the stream the kernels run on (HIP events)
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
        eng.stage_l0_input(feat)
        stages["l0_input"] = time_gpu(lambda: eng.stage_l0_input(feat), iters)
        stages["l0_conv1"] = time_gpu(eng.stage_l0_conv1, iters)
        stages["l0_conv2"] = time_gpu(lambda: eng.stage_l0_conv2(sample=False), iters)
        if use_image:
            stages["l0_sample1"] = time_gpu(lambda: (eng.stage_l0_conv2(sample=True)), iters) - stages["l0_conv2"]
        stages["pool1"] = time_gpu(eng.stage_pool1, iters)
        stages["tail"] = time_gpu(eng.stage_tail, iters)
        stages["head"] = time_gpu(lambda: eng._decode(eng.stage_head()), iters)
        kernels = {k: dict(ms=round(v, 4), alg_MB=round(ab[k] / 1e6, 2) if k in ab else None,
                           alg_GBs=round(ab[k] / 1e9 / (v / 1e3), 1) if k in ab else None)
                   for k, v in stages.items()}
        dom = max(("l0_conv1", "l0_conv2"), key=lambda k: stages[k])  # single-launch stages
        achieved = ab[dom] / 1e9 / (stages[dom] / 1e3)
        c0 = 19 if use_image else 3
        nt = eng.ntaps0
        kname = {"l0_conv1": (f"k_conv_l0_mixed<{c0 - 16}, {nt}>" if use_image else f"k_conv_l0_narrow<{c0}, {nt}>"),
                 "l0_conv2": (f"k_conv_l0_mfma<{c0}, {nt}>" if os.environ.get("DAGR_L0_MFMA", "1") != "0"
                              else f"k_conv_l0<16, {c0}, {nt}>")}[dom]
        # HBM bytes per launch from the PMC passes of this same command (tools/pmc.sh -> profiles/r1_traffic.json);
        # PMC counters cannot be read from inside this process
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r1_traffic.json")
        if NPW == 100000 and B == 8 and (W, H) == (640, 480) and a.stream == "uniform" and os.path.exists(tj):
            cfg = json.load(open(tj))["configs"]["use_image" if use_image else "events_only"]
            traffic = cfg.get(kname, {}).get("traffic_bytes")
        roofline = dict(kernel=kname, bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                        alg_bytes_per_launch=int(ab[dom]), launch_ms=round(stages[dom], 4))
        result = {
            "metric": "events_per_sec", "value": round(value, 1), "unit": "events/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("dagr-s + --use_image --img_net " + a.img_net if use_image else "dagr-s events-only")
                                   + f", {W}x{H} synthetic S-{a.stream}, B={B} windows/step x {NPW} events (50 ms "
                                   f"each), r={r}, K=16, events->graph->GNN(+image fusion)->decoded head outputs",
                       "image_branch": (None if not use_image else "in-line" if not a.pipeline_image
                                        else "one step ahead on a side stream"),
                       "engines": n_eng,
                       "events_per_step_per_gpu": n_events_step, "edges_per_step": int(ne),
                       "level_nodes_edges": levels,
                       # one batch through one engine, stages run back to back (sum of the stage timings below)
                       "batch_latency_ms": round(sum(stages.values()), 4)},
            "roofline": roofline, "stages": kernels,
        }
        if world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(model_cpu, sd_cpu, W, H, NPW, a.cpu_windows, a.stream, use_image,
                                                  a.img_net)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
