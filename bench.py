#!/usr/bin/env python3
"""Headline benchmark: images/s @640x640, batch 32, YOLOv8s-seg (BASELINE.json metric, configs[1]).

A "step" = one pass of the hot path over one device-resident batch of 32 synthetic 640x640 B-scans:
forward (75 convs + ConvT + pools + decode) + batched NMS + mask assembly, all on the HIP kernels behind
the C-ABI.  Inputs are resident in HBM before the timed region.  One process per GPU; inference shards
the image batch across ranks with NO data-path collective (weak scaling: 32 images per GPU per step);
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.

The input rotates over 8 distinct device-resident batches (315 MB > the 256 MiB Infinity Cache), so every step reads its
images from HBM.  `value` is PIPELINED THROUGHPUT: two batches are in flight on two engine instances and the
post-processing of a batch overlaps later forwards, so `ms_per_step` is the step period, not a latency.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured live with HIP events recorded
on the launch stream inside the timed region), at N=1 `cpu_baseline` (the CPU oracle, kind "port",
timed on this node's host cores on a bounded sample of the same workload, batch 32) and `train_step`: the training
path's step on a synthetic device-resident batch after the inference measurement -- N=1: YOLOv8s-seg, batch 64 @640
(BASELINE config 3) with the forward / loss / backward / optimizer split; N>1: YOLOv8m-seg data parallel, 64 images per
GPU (config 4), with the gradient all-reduce time alone, the part of it not hidden under backward and the overlap fraction.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--scale", default="s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the train_step measurement")
    ap.add_argument("--input-batches", type=int, default=8,
                    help="distinct device-resident input batches the steps rotate over (8 x 39 MB > the 256 MiB Infinity Cache)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-op HIP events")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run post-processing on the forward stream instead of overlapping it with the next batch")
    ap.add_argument("--engines", type=int, default=2,
                    help="engine instances (own activations and streams) that take the batches in turn: two batches "
                         "are in flight, the small-map tail of one forward runs beside the large-map start of the next")
    ap.add_argument("--lanes", choices=("auto", "on", "off"), default="auto",
                    help="stream lanes inside an engine (Proto / head levels on side streams): they gain 4-5 %% with one "
                         "engine and cost 3 %% once two engines are in flight, so auto = on for one engine, off otherwise")
    ap.add_argument("--serial", action="store_true",
                    help="one engine, one stream lane inside it (M355_NO_LANES), whole-batch launches (M355_NO_SUBBATCH): every kernel runs alone, so rocprofv3's "
                         "per-kernel averages and the live event samples describe the same launches")
    ap.add_argument("--profile-steps", type=int, default=4,
                    help="event-sampled steps (forward alone, a HIP event pair per launch) run AFTER the timed region; at least 2")
    ap.add_argument("--profile-every", type=int, default=100, help="(ignored since round 4: no step of the timed region is event-sampled)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))     # plain `python bench.py --gpus N`: one child launcher, before any GPU call
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # M355_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("M355_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.synthetic import synthetic_bscans
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict

    B = args.batch
    sd = synthetic_state_dict(args.scale, 1, seed=0)
    if args.serial:
        os.environ["M355_NO_LANES"] = "1"
        os.environ["M355_NO_SUBBATCH"] = "1"
        args.engines = 1
        args.no_overlap = True      # post-processing on the forward stream too: nothing runs beside any kernel
    n_eng = max(1, args.engines)
    if args.lanes == "off" or (args.lanes == "auto" and n_eng > 1):
        os.environ["M355_NO_LANES"] = "1"
    engs = []
    for _ in range(n_eng):
        e_ = SegEngine(args.scale, 1, (640, 640), max_batch=B, device=local_rank, keep_raw=False)   # as YOLO.predict does
        e_.load_state_dict(sd)
        engs.append(e_)
    eng = engs[0]
    n_in = max(1, args.input_batches)
    imgs_all = [torch.from_numpy(synthetic_bscans(B, seed=1000 + 97 * rank + j)).cuda() for j in range(n_in)]
    conf, iou, max_det = 0.25, 0.7, 300

    # persistent output buffers (caller-owned), allocated once outside the timed region
    import ctypes as C
    from defectdetection_viaobjectdetection_amd._capi import check, lib
    # Pipeline: batch i goes to engine i % n_eng, which owns a forward stream, a post-processing stream and two sets of
    # caller-owned output buffers.  The post-processing of a batch (NMS: one block per image, then the mask kernel)
    # runs beside later forwards; with two engines two forwards are in flight as well.  Every batch is fully processed
    # inside the timed region; --no-overlap puts post-processing on the forward stream, --serial removes all overlap.
    nbuf = 1 if args.no_overlap else 2
    mk = lambda shape, dt: [[torch.empty(shape, dtype=dt, device="cuda") for _ in range(nbuf)] for _ in range(n_eng)]  # noqa: E731
    preds = mk((B, eng.num_anchors, eng.pred_width), torch.float32)
    protos = mk((B, eng.proto_hw[0], eng.proto_hw[1], 32), torch.float16)
    dets = mk((B, max_det, 38), torch.float32)
    counts_b = [[torch.zeros((B,), dtype=torch.int32, device="cuda") for _ in range(nbuf)] for _ in range(n_eng)]
    masks = mk((B, max_det, 640, 640), torch.uint8)
    s_fwd = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n_eng - 1)]
    s_post = [s_fwd[e] if args.no_overlap else torch.cuda.Stream() for e in range(n_eng)]
    ev_fwd = [[torch.cuda.Event() for _ in range(nbuf)] for _ in range(n_eng)]
    ev_post = [[torch.cuda.Event() for _ in range(nbuf)] for _ in range(n_eng)]
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    step_no = [0]
    last_fwd = [None]     # event after the most recent forward (any engine)
    last_post = [None] * n_eng   # event after the most recent post-processing of each engine

    def step(alone=False):
        """alone: this forward overlaps nothing -- no other forward and no post-processing of an earlier batch (event-sampled
        steps: per-kernel times are clean).  A persistent kernel that needs a whole CU's LDS is not even dispatched while
        the mask kernel's small blocks keep refilling the CUs: measured on the model.1 patch kernel, 90 us from its first
        wave's entry to its last wave's exit (s_memrealtime) but 180 us between its events."""
        i = step_no[0]
        step_no[0] += 1
        e, k = i % n_eng, (i // n_eng) % nbuf
        sf, sp = s_fwd[e], s_post[e]
        if nbuf == 2:
            sf.wait_event(ev_post[e][k])          # the post-processing that last read this buffer set is done
        if (alone or step.prev_alone) and last_fwd[0] is not None:
            sf.wait_event(last_fwd[0])            # serialise against the previous forward (other engine's stream)
        if alone:
            for ev in last_post:
                if ev is not None:
                    sf.wait_event(ev)
        check(lib.m355_forward(engs[e]._h, P(imgs_all[i % n_in]), B, P(preds[e][k]), P(protos[e][k]), C.c_void_p(sf.cuda_stream)), engs[e]._h)
        ev_fwd[e][k].record(sf)
        last_fwd[0] = ev_fwd[e][k]
        step.prev_alone = alone
        if sp is not sf:
            sp.wait_event(ev_fwd[e][k])
        check(lib.m355_postprocess(engs[e]._h, P(preds[e][k]), P(protos[e][k]), B, conf, iou, max_det, P(dets[e][k]),
                                   P(counts_b[e][k]), P(masks[e][k]), C.c_void_p(sp.cuda_stream)), engs[e]._h)
        if nbuf == 2:
            ev_post[e][k].record(sp)
            last_post[e] = ev_post[e][k]
    step.prev_alone = False

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    profile = not args.no_profile
    if profile:
        for e_ in engs:
            e_.collect_op_times()  # drain + reset
    # The timed region is the plain loop: K steps, nothing else.  (Round 3 ran the event-sampled steps INSIDE it; a sampled step
    # drains and refills the pipeline, ~2.5 ms, so the figure depended on --steps: 6.5 % of a 20-step run.)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # Event-sampled steps, AFTER the clock has been read: each runs its forward alone (no other forward, no post-processing beside
    # it) with a HIP event pair around every launch, on the stream the kernels are launched on.
    sampled = 0
    if profile:
        for i in range(max(2, args.profile_steps)):
            cur = engs[step_no[0] % n_eng]
            cur.set_profiling(True)
            step(alone=True)
            cur.set_profiling(False)
            sampled += 1
        torch.cuda.synchronize()
    ms_sum, cnt = [], []
    if profile:
        for e_ in engs:   # every engine has the same op list: add the samples up
            m_, c_ = e_.collect_op_times()
            ms_sum = list(m_) if not ms_sum else [a + b for a, b in zip(ms_sum, m_)]
            cnt = list(c_) if not cnt else [a + b for a, b in zip(cnt, c_)]

    total_images = world * B * args.steps
    value = total_images / elapsed
    out = {
        "metric": "images/sec @640x640 b32 YOLOv8s-seg" if args.scale == "s" else f"images/sec @640x640 YOLOv8{args.scale}-seg",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "fp16", "data": "synthetic",
        "config": {"workload": f"YOLOv8{args.scale}-seg inference (forward + NMS + masks), {B} synthetic 640x640 "
                               f"B-scans per GPU per step, nc=1, seeded synthetic weights; pipelined throughput, "
                               f"{n_eng} batch(es) in flight (ms_per_step is the step period, not a latency); inputs rotate "
                               f"over {n_in} HBM-resident batches ({n_in * B * 640 * 640 * 3 / 2 ** 20:.0f} MiB)",
                   "input_batches": n_in,
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"batch-sharded x{world}, no collective",
                   "conf": conf, "iou": iou, "max_det": max_det,
                   "mean_detections_per_image": round(float(counts_b[0][0].float().mean()), 2),
                   "engines_in_flight": n_eng, "stream_lanes_per_engine": 1 if os.environ.get("M355_NO_LANES") else 3,
                   "pipelining": ("none" if args.no_overlap and n_eng == 1 else
                                  f"{n_eng} batch(es) in flight on {n_eng} engine instance(s); post-processing of a batch "
                                  f"overlaps later forwards (own HIP stream); event-sampled steps run their forward alone (no other forward, no post-processing beside it)")},
    }
    if rank == 0:
        gflop_img = eng.flops_per_image / 1e9
        out["config"]["conv_gflop_per_image"] = round(gflop_img, 3)
        out["config"]["whole_net_tflops"] = round(value / world * gflop_img / 1e3, 2)
        if profile:
            infos = eng.op_infos()
            by_kernel = {}
            for info, ms, c in zip(infos, ms_sum, cnt):
                if c == 0:
                    continue            # an op computed inside a neighbour's launch (the stem inside the model.1 patch kernel)
                k = by_kernel.setdefault(info["kernel"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
                k["ms"] += ms
                k["launches"] += c
                k["flops"] += info["flops"] * B * c
                k["bytes"] += (info["bytes"] * B + info["weight_bytes"]) * c
            dom = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
            d = by_kernel[dom]
            fwd_ms = sum(ms_sum) / max(sampled, 1)
            # which roof bounds the kernel: its arithmetic intensity (algorithmic FLOPs / algorithmic HBM bytes) against
            # the ridge of the two peaks.  1x1 convolutions sit far left of it (HBM), the 3x3 kernels right (MFMA).
            ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
            intensity = d["flops"] / max(d["bytes"], 1.0)
            if d["flops"] > 0 and intensity >= ridge:
                achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
                roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None}
            else:
                achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None}
            roof["flop_per_byte"] = round(intensity, 1)
            roof["traffic"], roof["traffic_source"] = pmc_traffic(dom)
            roof["launches_per_step"] = d["launches"] // max(sampled, 1)
            roof["event_sampled_steps"] = sampled
            roof["avg_launch_us"] = round(1e3 * d["ms"] / max(d["launches"], 1), 2)
            roof["algorithmic_gflop_per_launch"] = round(d["flops"] / max(d["launches"], 1) / 1e9, 3)
            roof["algorithmic_mbytes_per_launch"] = round(d["bytes"] / max(d["launches"], 1) / 1e6, 3)
            roof["share_of_forward_ms"] = round(d["ms"] / max(sum(ms_sum), 1e-9), 3)
            out["roofline"] = roof
            out["kernels"] = {k: {"ms_per_step": round(v["ms"] / max(sampled, 1), 4), "launches_per_step": v["launches"] // max(sampled, 1),
                                  "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 else 0.0,
                                  "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0}
                              for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])}
            out["forward_ms_per_step_events"] = round(fwd_ms, 4)
            # the north-star set: the 3x3 convolutions of the C2f bottlenecks (20 layers at the s scale: SURVEY 8d, 9.44 GFLOP per
            # image).  Sum of their algorithmic FLOPs over the sum of the event times of the launches that compute them; a fused
            # launch (a whole Bottleneck; model.2's block with its 1x1) is counted with its whole time and its 3x3 FLOPs only.
            cf = {"flops": 0.0, "ms": 0.0, "launches": 0, "layers": 0}
            for info, ms, c in zip(infos, ms_sum, cnt):
                if c == 0 or ".m." not in info["layer"]:
                    continue
                fused_block = info["kernel"].startswith("c2f_c32")          # two 3x3 (32 -> 32) + the 1x1 (96 -> 64): 18432 of 24576 MACs per pixel
                share = 18432.0 / 24576.0 if fused_block else 1.0
                cf["ms"] += ms
                cf["launches"] += c
                cf["layers"] += (2 if ("+cv2" in info["layer"]) else 1) * c
                cf["flops"] += share * info["flops"] * B * c
            if cf["ms"] > 0:
                tf = cf["flops"] / (cf["ms"] * 1e-3) / 1e12
                out["c2f3x3"] = {"tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / MFMA_PEAK_TFLOPS, 4), "peak": MFMA_PEAK_TFLOPS,
                                 "ms_per_step": round(cf["ms"] / max(sampled, 1), 4), "launches_per_step": cf["launches"] // max(sampled, 1),
                                 "layers": cf["layers"] // max(sampled, 1), "gflop_per_image": round(cf["flops"] / max(sampled, 1) / B / 1e9, 3),
                                 "target_frac": 0.70}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.scale, sd)
    if not args.no_train:
        for e_ in engs:
            e_.close()
        del preds, protos, dets, masks, imgs_all
        torch.cuda.empty_cache()
        try:
            ts = train_step_bench(world, dist)
        except Exception as ex:  # noqa: BLE001 -- the inference line must still be printed
            ts = {"error": f"{type(ex).__name__}: {ex}"}
        if rank == 0:
            out["train_step"] = ts
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher environment: start `python -m torch.distributed.run --nproc-per-node N
    bench.py <same arguments>` as a CHILD process (nothing in this process has touched the GPU, and it never will), let
    its output through (rank 0 prints the JSON line) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def train_step_bench(world, dist, steps=4, warm=2):
    """One optimizer step of the training path on a synthetic device-resident batch (images + 2 boxes / masks per image).
    N=1: YOLOv8s-seg, batch 64 @640 (BASELINE config 3); N>1: YOLOv8m-seg, 64 images per GPU, data parallel (config 4).
    Wall-clock between device synchronisations; every rank runs it, rank 0 reports (max over ranks for N>1)."""
    import ctypes as C
    import numpy as np
    import torch
    from defectdetection_viaobjectdetection_amd._capi import check, lib
    from defectdetection_viaobjectdetection_amd.loss import SegCriterion
    from defectdetection_viaobjectdetection_amd.sharding import GradBucketReducer
    from defectdetection_viaobjectdetection_amd.spec import init_state_dict
    from defectdetection_viaobjectdetection_amd.train_engine import TrainEngine
    scale, B, S = ("s", 64, 640) if world == 1 else ("m", 64, 640)
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = TrainEngine(scale, 1, (S, S), B, device=dev.index)
    eng.load_state_dict(init_state_dict(scale, 1, seed=0))
    rng = np.random.default_rng(int(os.environ.get("RANK", "0")))
    imgs = torch.from_numpy(rng.integers(0, 255, (B, S, S, 3), dtype=np.uint8)).to(dev)
    n = 2 * B
    boxes = torch.tensor(np.stack([rng.uniform(.3, .7, n), rng.uniform(.3, .7, n), rng.uniform(.1, .3, n), rng.uniform(.1, .3, n)], 1),
                         dtype=torch.float32).to(dev)
    masks = torch.zeros(B, S // 4, S // 4, device=dev)
    masks[:, 40:80, 40:80] = 1
    masks[:, 60:70, 60:70] = 2
    # labels as a loader hands them over: the small per-instance tensors on the host, the mask maps already on the device
    batch = {"batch_idx": torch.arange(B).repeat_interleave(2).float(), "cls": torch.zeros(n), "bboxes": boxes.cpu(), "masks": masks}
    criterion = SegCriterion(1, (S, S))
    m1 = torch.zeros(eng.n_train, device=dev)
    m2 = torch.zeros(eng.n_train, device=dev)
    ema = eng.flat_params.clone()
    reducer = GradBucketReducer(eng.flat_grads, eng.grad_spans()) if world > 1 else None
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731

    def tick():
        torch.cuda.synchronize()
        return time.perf_counter()

    def one_step(it, tim, overlap=True, comm=True):
        """One step as a training loop runs it: the host synchronises ONCE, at the end (reading the loss); the phase times are
        device times between events on the stream (an idle device waiting for the host counts in the phase it waits in)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        prep = criterion.prepare(batch, B, dev)                           # padded targets before the forward is enqueued
        ev[0].record()
        raw, protos = eng.forward(imgs)
        ev[1].record()
        items, d_raw, d_protos = criterion(raw, protos, prep, 128.0)      # loss + backward of the loss
        ev[2].record()
        if reducer is not None and comm:
            reducer.reset()
        # N > 1: the pass without the exchange reports through a no-op callback, so that both passes enqueue the backward the same
        # way (with a callback attached TrainEngine keeps the head's backward on one stream) and their difference is the exchange
        cb = None if reducer is None else (reducer.mark_ready if (comm and overlap) else (lambda name: None))
        eng.backward(d_raw, d_protos, on_ready=cb)
        if reducer is not None and comm:
            reducer.finish()
        ev[3].record()
        check(lib.m355_adamw_step(eng.flat_params.data_ptr(), eng.flat_grads.data_ptr(), m1.data_ptr(), m2.data_ptr(), ema.data_ptr(),
                                  eng.group.data_ptr(), eng.n_train, 1e-4, 1e-4, 0.9, 0.999, 1e-8, 5e-4, it + 1, 1 / (128.0 * world), 0.999,
                                  st()))
        ev[4].record()
        eng.repack()
        ev[5].record()
        loss = float(items.sum() * B)                                     # the step's host synchronisation
        ev[5].synchronize()
        for j, k in enumerate(("fwd", "loss", "bwd", "opt", "repack")):
            tim[k] = tim.get(k, 0.0) + ev[j].elapsed_time(ev[j + 1]) * 1e-3
        return loss

    def run(overlap, comm):
        tim = {}
        for it in range(warm):
            one_step(it, {}, overlap, comm)
        if dist is not None:
            dist.barrier()
        t0 = tick()
        for it in range(steps):
            loss = one_step(warm + it, tim, overlap, comm)
        total = tick() - t0
        return total / steps, {k: v / steps for k, v in tim.items()}, loss

    per_step, tim, loss = run(True, True)
    res = {"config": f"YOLOv8{scale}-seg training step, {B} synthetic {S}x{S} images per GPU, nc=1, fresh weights, AdamW; "
                     f"{'single GPU' if world == 1 else f'data parallel x{world}, SUM all-reduce of the flat fp32 gradient in 32 MiB buckets under backward'}",
           "steps": steps, "warmup": warm, "ms_per_step": round(per_step * 1e3, 2), "images_per_s": round(world * B / per_step, 1),
           # forward conv FLOPs per image @640, nc=1; a training step ~ 3x (forward + dgrad + wgrad)
           "tflops": round(3 * {"n": 11.34, "s": 39.92, "m": 104.28}[scale] * 1e9 * B / per_step / 1e12, 1),
           "forward_ms": round(tim["fwd"] * 1e3, 2), "loss_ms": round(tim["loss"] * 1e3, 2), "backward_ms": round(tim["bwd"] * 1e3, 2),
           "optimizer_ms": round(tim["opt"] * 1e3, 2), "repack_ms": round(tim["repack"] * 1e3, 2), "loss": round(loss, 4)}
    if world > 1:
        _, tim_nc, _ = run(False, False)                     # backward without any exchange
        torch.cuda.synchronize()
        dist.barrier()
        t0 = tick()
        for _ in range(steps):                               # the exchange alone: one SUM all-reduce of the whole flat buffer
            dist.all_reduce(eng.flat_grads, op=dist.ReduceOp.SUM)
        comm_alone = (tick() - t0) / steps
        exposed = max(tim["bwd"] - tim_nc["bwd"], 0.0)
        vals = torch.tensor([per_step, comm_alone, exposed, tim["bwd"], tim_nc["bwd"]], dtype=torch.float64, device=dev)
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        per_step, comm_alone, exposed, bwd_c, bwd_nc = [float(v) for v in vals]
        res.update({"ms_per_step": round(per_step * 1e3, 2), "images_per_s": round(world * B / per_step, 1),
                    "gradient_mbytes": round(eng.n_train * 4 / 1e6, 1), "allreduce_ms": round(comm_alone * 1e3, 3),
                    "allreduce_exposed_ms": round(exposed * 1e3, 3), "backward_ms": round(bwd_c * 1e3, 2),
                    "backward_no_exchange_ms": round(bwd_nc * 1e3, 2),
                    "overlap_frac": round(min(max(1.0 - exposed / max(comm_alone, 1e-9), 0.0), 1.0), 3),
                    "buckets": len(reducer.launched)})
    del eng
    torch.cuda.empty_cache()
    return res


def pmc_traffic(kernel_label):
    """(HBM bytes per launch of `kernel_label`, the file they come from): the newest committed rocprofv3 PMC pass of this
    same command under profiles/ (rNN_pmc_traffic.json; FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE).  Not
    measured in this run -- counters need their own rocprofv3 passes -- so the line names its source; (None, None) if the
    newest pass does not hold the kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                k = json.load(f)["kernels"].get(kernel_label)
        except (OSError, ValueError, KeyError):
            continue
        if k is not None:
            return round(k["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)
    return None, None


def host_cores():
    """CPU threads this process may really use: the cgroup quota if there is one, else the affinity mask,
    capped at 16 (a one-GPU box's CPU share on this pool) so the baseline is not oversubscribed."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def cpu_baseline(scale, sd):
    """The CPU oracle (a restatement of the Ultralytics CPU path; kind "port") on this node's host cores,
    bounded sample: batches of 32 of the same synthetic workload, 1 warm-up + timed iterations for ~15-20 s."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import yolov8_seg_oracle as orc
    from defectdetection_viaobjectdetection_amd.synthetic import synthetic_bscans
    cores = host_cores()
    torch.set_num_threads(cores)
    model = orc.SegmentationModel(scale, 1)
    model.load_state_dict(sd)
    model.eval()
    bs = 32
    imgs = synthetic_bscans(bs, seed=1000)
    x = torch.from_numpy(imgs.transpose(0, 3, 1, 2).copy()).float() / 255.0

    def one():
        with torch.no_grad():
            preds, protos = model(x)
        dets = orc.non_max_suppression(preds.numpy(), 1, 0.25, 0.7, 300)
        for i, d in enumerate(dets):
            orc.process_mask(protos[i], torch.from_numpy(d[:, 6:]), torch.from_numpy(d[:, :4]), (640, 640))

    one()
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > 15.0 or n >= 12:
            break
    return {"value": round(n * bs / el, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} iterations of batch {bs} (640x640 synthetic B-scans) after 1 warm-up, PyTorch-CPU fp32 "
                      f"forward + numpy NMS + mask assembly, {cores} threads"}


if __name__ == "__main__":
    main()
