#!/usr/bin/env python3
"""Headline benchmark: images/s @640x640, batch 32, YOLOv8s-seg (BASELINE.json metric, configs[1]).

A "step" = one pass of the hot path over one device-resident batch of 32 synthetic 640x640 B-scans:
forward (75 convs + ConvT + pools + decode) + batched NMS + mask assembly, all on the HIP kernels behind
the C-ABI.  Inputs are resident in HBM before the timed region.  One process per GPU; inference shards
the image batch across ranks with NO data-path collective (weak scaling: 32 images per GPU per step);
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured live with HIP events recorded
on the launch stream inside the timed region) and, at N=1, `cpu_baseline` (the CPU oracle, kind "port",
timed on this node's host cores on a bounded sample of the same workload).
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
    ap.add_argument("--profile-every", type=int, default=25,
                    help="record per-op HIP events on every n-th timed step (an event pair per launch costs "
                         "~8 us of serialisation, ~0.6 ms per fully instrumented step)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
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

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import synthetic_bscans
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict

    B = args.batch
    sd = synthetic_state_dict(args.scale, 1, seed=0)
    if args.serial:
        os.environ["M355_NO_LANES"] = "1"
        os.environ["M355_NO_SUBBATCH"] = "1"
        args.engines = 1
    n_eng = max(1, args.engines)
    if args.lanes == "off" or (args.lanes == "auto" and n_eng > 1):
        os.environ["M355_NO_LANES"] = "1"
    engs = []
    for _ in range(n_eng):
        e_ = SegEngine(args.scale, 1, (640, 640), max_batch=B, device=local_rank, keep_raw=False)   # as YOLO.predict does
        e_.load_state_dict(sd)
        engs.append(e_)
    eng = engs[0]
    imgs = torch.from_numpy(synthetic_bscans(B, seed=1000 + rank)).cuda()
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

    def step(alone=False):
        """alone: this forward does not overlap any other forward (event-sampled steps: per-kernel times are clean)."""
        i = step_no[0]
        step_no[0] += 1
        e, k = i % n_eng, (i // n_eng) % nbuf
        sf, sp = s_fwd[e], s_post[e]
        if nbuf == 2:
            sf.wait_event(ev_post[e][k])          # the post-processing that last read this buffer set is done
        if (alone or step.prev_alone) and last_fwd[0] is not None:
            sf.wait_event(last_fwd[0])            # serialise against the previous forward (other engine's stream)
        check(lib.m355_forward(engs[e]._h, P(imgs), B, P(preds[e][k]), P(protos[e][k]), C.c_void_p(sf.cuda_stream)), engs[e]._h)
        ev_fwd[e][k].record(sf)
        last_fwd[0] = ev_fwd[e][k]
        step.prev_alone = alone
        if sp is not sf:
            sp.wait_event(ev_fwd[e][k])
        check(lib.m355_postprocess(engs[e]._h, P(preds[e][k]), P(protos[e][k]), B, conf, iou, max_det, P(dets[e][k]),
                                   P(counts_b[e][k]), P(masks[e][k]), C.c_void_p(sp.cuda_stream)), engs[e]._h)
        if nbuf == 2:
            ev_post[e][k].record(sp)
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
    sampled = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        # every n-th step; a run shorter than n samples its middle step only (sampled steps give up all overlap)
        on = profile and (i % args.profile_every == args.profile_every // 2 or
                          (args.steps < args.profile_every and i == args.steps // 2))
        cur = engs[step_no[0] % n_eng]
        if on:
            cur.set_profiling(True)
            sampled += 1
        step(alone=on)
        if on:
            cur.set_profiling(False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
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
                               f"B-scans per GPU per step, nc=1, seeded synthetic weights",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"batch-sharded x{world}, no collective",
                   "conf": conf, "iou": iou, "max_det": max_det,
                   "mean_detections_per_image": round(float(counts_b[0][0].float().mean()), 2),
                   "engines_in_flight": n_eng, "stream_lanes_per_engine": 1 if os.environ.get("M355_NO_LANES") else 3,
                   "pipelining": ("none" if args.no_overlap and n_eng == 1 else
                                  f"{n_eng} batch(es) in flight on {n_eng} engine instance(s); post-processing of a batch "
                                  f"overlaps later forwards (own HIP stream); event-sampled steps run their forward alone")},
    }
    if rank == 0:
        gflop_img = eng.flops_per_image / 1e9
        out["config"]["conv_gflop_per_image"] = round(gflop_img, 3)
        out["config"]["whole_net_tflops"] = round(value / world * gflop_img / 1e3, 2)
        if profile:
            infos = eng.op_infos()
            by_kernel = {}
            for info, ms, c in zip(infos, ms_sum, cnt):
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
            roof["traffic"] = pmc_traffic(dom)
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
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.scale, sd)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def pmc_traffic(kernel_label):
    """HBM bytes per launch of `kernel_label` from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_pmc_traffic.json; FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), or None."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            k = json.load(f)["kernels"].get(kernel_label)
        return None if k is None else round(k["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        return None


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
    bounded sample: batches of 8 of the same synthetic workload, 1 warm-up + timed iterations for ~15 s."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import yolov8_seg_oracle as orc
    from helpers import synthetic_bscans
    cores = host_cores()
    torch.set_num_threads(cores)
    model = orc.SegmentationModel(scale, 1)
    model.load_state_dict(sd)
    model.eval()
    bs = 8
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
        if el > 15.0 or n >= 40:
            break
    return {"value": round(n * bs / el, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} iterations of batch {bs} (640x640 synthetic B-scans) after 1 warm-up, PyTorch-CPU fp32 "
                      f"forward + numpy NMS + mask assembly, {cores} threads"}


if __name__ == "__main__":
    main()
