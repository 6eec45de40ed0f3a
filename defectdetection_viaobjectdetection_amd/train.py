"""Training / validation entry points behind ``YOLO.train`` and ``YOLO.val``.

Stands where upstream's ``SegmentationTrainer`` / ``SegmentationValidator`` stand for the call
``model.train(data="data-seg.yaml", epochs=30, imgsz=320, project=..., name=..., device=0)``
(/root/reference/BscanBased/yolo_seg_train.py:12-19): same keyword names, same run-directory contract
(``<project>/<name>/weights/{last,best}.pt`` + ``results.csv``, ``:10-11``), same trainer defaults (A14).

One step = ``TrainEngine.forward`` (HIP conv + batch-stat BN kernels) -> loss on the head outputs (``loss.py``) ->
``TrainEngine.backward`` (HIP BN-bwd / wgrad / dgrad kernels) -> SUM all-reduce of the flat gradient buffer over RCCL
in buckets overlapped with backward (N > 1 only) -> one fused optimizer + EMA kernel over the flat fp32 parameters.
fp16 activations / gradients with a dynamic loss scale play the role of upstream's AMP GradScaler.  There is no CPU
path: without a GPU and ``libmi355yolo.so`` this raises.

Augmentation (mosaic, scale / translate affine, HSV, flip) is one GPU gather kernel over the HBM-resident image cache
(``augment.py`` / ``csrc/augment.hip``).  ``resume=True`` continues from ``weights/last.pt`` (optimizer, EMA, loss scale,
history); ``patience`` stops early.  Not built: plots.
"""
from __future__ import annotations

import csv
import math
import os
import subprocess
import sys
import time
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import metrics as M
from ._capi import check, lib
from .dataset import SegDataset, epoch_batches, rasterize_polygon, read_data_yaml
from .loss import SegCriterion
from .sharding import GradBucketReducer

DEFAULTS = dict(optimizer="auto", lr0=0.01, lrf=0.01, momentum=0.937, weight_decay=5e-4, warmup_epochs=3.0,
                warmup_momentum=0.8, warmup_bias_lr=0.1, nbs=64, seed=0, fliplr=0.5, val=True, exist_ok=False,
                box=7.5, cls=0.5, dfl=1.5, conf=0.001, iou=0.7, max_det=300, workers=8, patience=100, amp=True,
                verbose=True, save=True, bucket_mb=32, augment=True, mosaic=1.0, scale=0.5, translate=0.1, hsv_h=0.015,
                hsv_s=0.7, hsv_v=0.4, close_mosaic=10)
EMA_DECAY, EMA_TAU = 0.9999, 2000.0
GRAD_CLIP = 10.0


def _devices(device) -> List[int]:
    if isinstance(device, (list, tuple)):
        return [int(d) for d in device]
    if isinstance(device, str):
        if device.lower() == "cpu":
            raise RuntimeError("training runs on the HIP kernels only; device='cpu' is not available")
        return [int(d) for d in device.replace("cuda:", "").split(",") if d.strip() != ""]
    return [int(device)]


def _run_dir(project: Optional[str], name: Optional[str], exist_ok: bool) -> str:
    base = os.path.join(project or os.path.join("runs", "segment"), name or "train")
    if exist_ok or not os.path.exists(base):
        return base
    i = 2
    while os.path.exists(f"{base}{i}"):
        i += 1
    return f"{base}{i}"


def lr_factor(epoch: int, epochs: int, lrf: float) -> float:
    return max(1.0 - epoch / epochs, 0.0) * (1.0 - lrf) + lrf


def ema_decay(updates: int) -> float:
    return EMA_DECAY * (1.0 - math.exp(-updates / EMA_TAU))


class LossScaler:
    """Dynamic loss scale for the fp16 backward pass: halve on a non-finite gradient, double after ``interval`` clean steps."""

    def __init__(self, init: float = 1024.0, interval: int = 200, lo: float = 1.0, hi: float = 65536.0):
        self.scale, self.interval, self.lo, self.hi, self._good = init, interval, lo, hi, 0

    def update(self, found_inf: bool) -> None:
        if found_inf:
            self.scale = max(self.scale * 0.5, self.lo)
            self._good = 0
        else:
            self._good += 1
            if self._good >= self.interval:
                self.scale = min(self.scale * 2.0, self.hi)
                self._good = 0


class Validator:
    """A17: NMS at conf 0.001 on the inference engine, box and mask IoU matching, 101-point AP."""

    def __init__(self, dataset: SegDataset, scale: str, nc: int, device: int, batch: int = 16, conf: float = 0.001,
                 iou: float = 0.7, max_det: int = 300):
        from .engine import SegEngine
        self.ds, self.nc, self.device = dataset, nc, device
        self.conf, self.iou, self.max_det = conf, iou, max_det
        self.batch = max(1, min(batch, len(dataset)))
        self.engine = SegEngine(scale, nc, dataset.imgsz, max_batch=self.batch, device=device, keep_raw=False)
        self._gt: Dict[int, Tuple[np.ndarray, np.ndarray, torch.Tensor]] = {}

    def _ground_truth(self, i: int):
        if i not in self._gt:
            H, W = self.ds.imgsz
            inst = self.ds.labels[i]
            cls = np.array([c for c, _ in inst], np.int64)
            boxes = np.array([[p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max()] for _, p in inst],
                             np.float64).reshape(-1, 4)
            masks = np.stack([rasterize_polygon(p, H, W) for _, p in inst]) if inst else np.zeros((0, H, W), bool)
            self._gt[i] = (cls, boxes, torch.from_numpy(masks))
        return self._gt[i]

    def __call__(self, state_dict: Dict[str, torch.Tensor]) -> Dict[str, float]:
        self.engine.load_state_dict(state_dict)
        dev = torch.device("cuda", self.device)
        tp_b, tp_m, confs, pcls, gcls = [], [], [], [], []
        n = len(self.ds)
        with torch.cuda.device(dev):
            for i0 in range(0, n, self.batch):
                idx = list(range(i0, min(i0 + self.batch, n)))
                x = torch.from_numpy(self.ds.images[idx]).to(dev)
                preds, protos = self.engine.forward(x)
                dets, counts, masks = self.engine.postprocess(preds, protos, self.conf, self.iou, self.max_det, masks=True,
                                                              multi_label=True)   # upstream's validator mode (a no-op for nc = 1)
                counts_h = counts.cpu().tolist()
                for j, i in enumerate(idx):
                    k = counts_h[j]
                    d = dets[j, :k, :6].float().cpu().numpy()
                    g_cls, g_boxes, g_masks = self._ground_truth(i)
                    gcls.append(g_cls)
                    if k == 0:
                        continue
                    iou_b = M.box_iou(d[:, :4], g_boxes)
                    if g_cls.size:
                        pm = masks[j, :k].flatten(1).float()
                        gm = g_masks.to(dev).flatten(1).float()
                        inter = pm @ gm.T
                        union = pm.sum(1)[:, None] + gm.sum(1)[None, :] - inter
                        iou_m = (inter / (union + 1e-7)).double().cpu().numpy()
                    else:
                        iou_m = np.zeros((k, 0))
                    c = d[:, 5].astype(np.int64)
                    tp_b.append(M.match(c, g_cls, iou_b))
                    tp_m.append(M.match(c, g_cls, iou_m))
                    confs.append(d[:, 4])
                    pcls.append(c)
        gt_all = np.concatenate(gcls) if gcls else np.zeros(0, np.int64)
        if confs:
            conf_all, cls_all = np.concatenate(confs), np.concatenate(pcls)
            rb = M.summarize(np.concatenate(tp_b), conf_all, cls_all, gt_all)
            rm = M.summarize(np.concatenate(tp_m), conf_all, cls_all, gt_all)
        else:
            rb = rm = (0.0, 0.0, 0.0, 0.0)
        out = {"metrics/precision(B)": rb[0], "metrics/recall(B)": rb[1], "metrics/mAP50(B)": rb[2], "metrics/mAP50-95(B)": rb[3],
               "metrics/precision(M)": rm[0], "metrics/recall(M)": rm[1], "metrics/mAP50(M)": rm[2], "metrics/mAP50-95(M)": rm[3]}
        out["fitness"] = 0.1 * (rb[2] + rm[2]) + 0.9 * (rb[3] + rm[3])
        return out

    def close(self) -> None:
        self.engine.close()


def _metrics_namespace(res: Dict[str, float], save_dir: Optional[str]) -> SimpleNamespace:
    box = SimpleNamespace(mp=res["metrics/precision(B)"], mr=res["metrics/recall(B)"], map50=res["metrics/mAP50(B)"],
                          map=res["metrics/mAP50-95(B)"])
    seg = SimpleNamespace(mp=res["metrics/precision(M)"], mr=res["metrics/recall(M)"], map50=res["metrics/mAP50(M)"],
                          map=res["metrics/mAP50-95(M)"])
    return SimpleNamespace(box=box, seg=seg, fitness=res["fitness"], results_dict=dict(res), save_dir=save_dir)


def validate(model, data=None, imgsz=None, batch=16, device=0, conf=0.001, iou=0.7, max_det=300, split="val", **kwargs):
    data = data or model.train_args.get("data")
    if not data:
        raise ValueError("val() needs data=<dataset yaml> (or a checkpoint trained by this package)")
    cfg = read_data_yaml(data)
    if cfg["nc"] != model.nc:
        raise ValueError(f"dataset has {cfg['nc']} classes, model has {model.nc}")
    imgsz = int(imgsz or model.train_args.get("imgsz", 640))
    ds = SegDataset(cfg[split], imgsz, nc=model.nc)
    v = Validator(ds, model.scale, model.nc, _devices(device)[0], batch, conf, iou, max_det)
    try:
        res = v(model.state_dict)
    finally:
        v.close()
    return _metrics_namespace(res, None)


def _spawn_ddp(model, devices: List[int], kwargs: Dict) -> str:
    """``device=[0,1,...]`` without a launcher: one process per GPU through torch.distributed.run (the way upstream
    generates a DDP script and re-launches itself).  Returns the run directory the children wrote."""
    import socket
    import tempfile
    save_dir = _run_dir(kwargs.get("project"), kwargs.get("name"), kwargs.get("exist_ok", False))
    os.makedirs(os.path.join(save_dir, "weights"), exist_ok=True)
    init = model.save(os.path.join(save_dir, "weights", "init.pt"))
    kw = dict(kwargs, device=devices, exist_ok=True, project=os.path.dirname(save_dir) or ".", name=os.path.basename(save_dir))
    with tempfile.NamedTemporaryFile("w", suffix="_ddp.py", delete=False, dir=save_dir) as f:
        f.write(f"import sys\nsys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
                f"from defectdetection_viaobjectdetection_amd.model import YOLO\nYOLO({init!r}).train(**{kw!r})\n")
        script = f.name
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={len(devices)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script]
    try:
        subprocess.run(cmd, check=True, env=env)
    finally:
        os.unlink(script)
    return save_dir


def train(model, data=None, epochs=100, imgsz=640, batch=16, project=None, name=None, device=0, **kwargs):
    unknown = [k for k in kwargs if k not in DEFAULTS and k not in ("pretrained", "task", "mode", "model", "resume", "max_steps")]
    if unknown:
        raise TypeError(f"train() got unexpected keyword(s) {unknown}; known: {sorted(DEFAULTS)}")
    resume_state = None
    if kwargs.get("resume"):
        resume_state = getattr(model, "_resume_state", None)
        if not resume_state:
            raise ValueError("resume=True needs a model loaded from a weights/last.pt written by this trainer")
        ra = model.train_args
        data = data or ra.get("data")
        epochs, imgsz, batch = int(ra.get("epochs", epochs)), int(ra.get("imgsz", imgsz)), int(ra.get("batch", batch))
        kwargs = {**{k: v for k, v in ra.get("options", {}).items() if k in DEFAULTS}, **{k: v for k, v in kwargs.items() if k != "resume"}}
    if not data:
        raise ValueError("train() needs data=<dataset yaml>")
    a = SimpleNamespace(**{**DEFAULTS, **{k: v for k, v in kwargs.items() if k in DEFAULTS}})
    devices = _devices(device)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if len(devices) > 1 and world == 1:
        run = _spawn_ddp(model, devices, dict(kwargs, data=data, epochs=epochs, imgsz=imgsz, batch=batch, project=project, name=name))
        best = os.path.join(run, "weights", "best.pt")
        model._load_checkpoint(best if os.path.isfile(best) else os.path.join(run, "weights", "last.pt"))
        model._drop_engines()
        return SimpleNamespace(save_dir=run)
    if not torch.cuda.is_available():
        raise RuntimeError("YOLO.train needs a gfx950 GPU: the training path is the HIP kernels, there is no CPU fallback")

    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = devices[local_rank] if world > 1 and len(devices) == world else (local_rank if world > 1 else devices[0])
    # Rehearsal on a one-GPU box: every rank on the same device and gloo as the transport (RCCL refuses two ranks on
    # one GPU).  Everything else -- batch sharding, bucketed all-reduce under backward, rank-0 bookkeeping -- is the
    # production path.
    backend = os.environ.get("M355_DIST_BACKEND", "nccl")
    if os.environ.get("M355_DIST_SAME_DEVICE"):
        dev_index = devices[0]
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    own_pg = False
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        own_pg = True
    if batch % world:
        raise ValueError(f"batch {batch} must be a multiple of the number of GPUs {world}")
    local_batch = batch // world

    cfg = read_data_yaml(data)
    if cfg["nc"] != model.nc:
        model.set_classes(cfg["nc"], cfg["names"])        # upstream re-heads the model when data.yaml disagrees
    else:
        model.names = dict(cfg["names"])
    imgsz = int(math.ceil(int(imgsz) / 32) * 32)
    train_ds = SegDataset(cfg["train"], imgsz, nc=model.nc)
    val_ds = SegDataset(cfg["val"], imgsz, nc=model.nc) if a.val and rank == 0 else None

    from .train_engine import TrainEngine
    eng = TrainEngine(model.scale, model.nc, (imgsz, imgsz), local_batch, device=dev_index)
    eng.load_state_dict(model.state_dict)
    n_train = eng.n_train
    flat_p, flat_g = eng.flat_params, eng.flat_grads
    ema = flat_p.clone()
    state1 = torch.zeros(n_train, device=dev)
    state2 = torch.zeros(n_train, device=dev)
    acc = torch.zeros(n_train, device=dev)
    sumsq = torch.zeros(int(lib.m355_grad_sumsq_workspace_floats()), device=dev)   # [0] sum g^2, [1] non-finite count, then workspace
    reducer = GradBucketReducer(flat_g, eng.grad_spans(), bucket_bytes=int(a.bucket_mb) << 20) if world > 1 else None
    criterion = SegCriterion(model.nc, (imgsz, imgsz), (a.box, a.cls, a.dfl))

    nb = len(epoch_batches(len(train_ds), local_batch, 0, a.seed, rank, world))
    opt_name = str(a.optimizer).lower()
    lr0, momentum, warmup_bias_lr = float(a.lr0), float(a.momentum), float(a.warmup_bias_lr)
    if opt_name == "auto":                                  # A14 "auto" rule
        iters = math.ceil(len(train_ds) / max(batch, a.nbs)) * epochs
        if iters > 10000:
            opt_name, lr0, momentum = "sgd", 0.01, 0.9
        else:
            opt_name, lr0, momentum = "adamw", round(0.002 * 5 / (4 + model.nc), 6), 0.9
        warmup_bias_lr = 0.0
    if opt_name not in ("sgd", "adamw", "adam"):
        raise ValueError(f"optimizer '{a.optimizer}' is not supported (auto | SGD | AdamW)")
    accumulate = max(round(a.nbs / batch), 1)
    weight_decay = float(a.weight_decay) * batch * accumulate / a.nbs
    nw = max(round(a.warmup_epochs * nb), 100) if a.warmup_epochs > 0 else -1
    max_steps = kwargs.get("max_steps")

    save_dir = resume_state["save_dir"] if resume_state else _run_dir(project, name, a.exist_ok or world > 1)
    wdir = os.path.join(save_dir, "weights")
    if rank == 0:
        os.makedirs(wdir, exist_ok=True)
    model.train_args = dict(data=os.path.abspath(data), epochs=epochs, imgsz=imgsz, batch=batch, optimizer=opt_name, lr0=lr0,
                            momentum=momentum, weight_decay=weight_decay, seed=a.seed,
                            options={k: getattr(a, k) for k in DEFAULTS if isinstance(getattr(a, k), (int, float, str, bool))})
    validator = Validator(val_ds, model.scale, model.nc, dev_index, 16, a.conf, a.iou, a.max_det) if val_ds is not None else None
    augmenter = None
    if a.augment:
        from .augment import Augmenter
        augmenter = Augmenter(train_ds, dev, seed=a.seed + 1000 * rank, mosaic=a.mosaic, scale=a.scale, translate=a.translate,
                              hsv_h=a.hsv_h, hsv_s=a.hsv_s, hsv_v=a.hsv_v, fliplr=a.fliplr)
    scaler = LossScaler()
    rng = np.random.default_rng(a.seed + 1000 * rank)
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    fields = ["epoch", "time", "train/box_loss", "train/seg_loss", "train/cls_loss", "train/dfl_loss",
              "metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)",
              "metrics/precision(M)", "metrics/recall(M)", "metrics/mAP50(M)", "metrics/mAP50-95(M)", "lr/pg0", "loss_scale"]
    rows: List[Dict] = []
    best_fit, updates, opt_steps, skipped, global_step = -1.0, 0, 0, 0, 0
    last_res: Dict[str, float] = {}
    t_start = time.time()
    if rank == 0 and a.verbose:
        print(f"train: yolov8{model.scale}-seg nc={model.nc} imgsz={imgsz} batch={batch} (x{world} GPU, {local_batch}/GPU) "
              f"{len(train_ds)} images, {nb} it/epoch, optimizer={opt_name} lr0={lr0} wd={weight_decay:g} accumulate={accumulate}")

    best_epoch, start_epoch = -1, 0
    if resume_state:                                   # optimizer / EMA / scaler state of the interrupted run
        flat_p.copy_(resume_state["flat_params"].to(dev))
        ema.copy_(resume_state["ema"].to(dev))
        state1.copy_(resume_state["state1"].to(dev))
        state2.copy_(resume_state["state2"].to(dev))
        eng.repack()
        opt_steps, updates = int(resume_state["opt_steps"]), int(resume_state["updates"])
        scaler.scale = float(resume_state["loss_scale"])
        best_fit, best_epoch = float(resume_state["best_fit"]), int(resume_state["best_epoch"])
        rows = list(resume_state["rows"])
        start_epoch = int(resume_state["epoch"])
        if rank == 0 and a.verbose:
            print(f"resuming {save_dir} from epoch {start_epoch + 1}/{epochs}")

    def ckpt(path: str, sd: Dict[str, torch.Tensor], epoch: int, with_trainer: bool = False) -> None:
        ck = {"format": "mi355yolo-seg-v1", "scale": model.scale, "nc": model.nc, "names": model.names,
              "train_args": model.train_args, "model": sd, "epoch": epoch, "metrics": last_res}
        if with_trainer:                               # what resume=True needs (last.pt only)
            ck["trainer"] = {"epoch": epoch, "flat_params": flat_p.cpu(), "ema": ema.cpu(), "state1": state1.cpu(),
                             "state2": state2.cpu(), "opt_steps": opt_steps, "updates": updates, "loss_scale": scaler.scale,
                             "best_fit": best_fit, "best_epoch": best_epoch, "rows": rows, "save_dir": save_dir}
        torch.save(ck, path)

    stop = False
    for epoch in range(start_epoch, epochs):
        lf = lr_factor(epoch, epochs, a.lrf)
        lr = lr_bias = lr0 * lf
        mom = momentum
        mloss = torch.zeros(4, device=dev)
        batches = epoch_batches(len(train_ds), local_batch, epoch, a.seed, rank, world)
        micro = 0
        for i, idx in enumerate(batches):
            ni = i + nb * epoch
            if ni <= nw:
                f = ni / nw
                accumulate = max(1, round(1 + f * (a.nbs / batch - 1)))
                lr = f * lr0 * lf
                lr_bias = warmup_bias_lr + f * (lr0 * lf - warmup_bias_lr)
                mom = a.warmup_momentum + f * (momentum - a.warmup_momentum) if opt_name == "sgd" else momentum
            if augmenter is not None:     # mosaic / affine / HSV / flip rendered by the GPU kernel from the HBM image cache
                b = augmenter.batch(idx, mosaic_on=epoch < epochs - a.close_mosaic)
                imgs = b["img"]
            else:
                b = train_ds.batch(idx, flip=rng.random(len(idx)) < a.fliplr)
                imgs = torch.from_numpy(b["img"]).to(dev, non_blocking=True)
            # targets padded on the host before the forward is enqueued: the loss then needs no synchronisation of its own
            labels = criterion.prepare({k: torch.from_numpy(b[k]) for k in ("batch_idx", "cls", "bboxes", "masks")}, eng.B, dev)
            raw, protos = eng.forward(imgs)
            items, d_raw, d_protos = criterion(raw, protos, labels, scaler.scale)     # loss + its backward
            micro += 1
            step_now = micro >= accumulate or i == len(batches) - 1
            overlap = reducer is not None and step_now and accumulate == 1
            if overlap:
                reducer.reset()
            eng.backward(d_raw, d_protos, on_ready=reducer.mark_ready if overlap else None)
            mloss += items
            if accumulate > 1 or not step_now:
                acc.add_(flat_g)
                if not step_now:
                    global_step += 1
                    continue
                flat_g.copy_(acc)
                acc.zero_()
            if reducer is not None:
                if not overlap:
                    reducer.reset()
                reducer.finish()
            micro = 0
            check(lib.m355_grad_sumsq(flat_g.data_ptr(), n_train, sumsq.data_ptr(), st()))
            ss, bad = sumsq[:2].tolist()
            found_inf = bad > 0 or not math.isfinite(ss)
            if not found_inf:
                gnorm = math.sqrt(ss) / scaler.scale
                grad_mul = min(1.0, GRAD_CLIP / (gnorm + 1e-6)) / scaler.scale
                updates += 1
                opt_steps += 1
                d = ema_decay(updates)
                if opt_name == "sgd":
                    check(lib.m355_sgd_step(flat_p.data_ptr(), flat_g.data_ptr(), state1.data_ptr(), ema.data_ptr(),
                                            eng.group.data_ptr(), n_train, lr, lr_bias, mom, 1, weight_decay, grad_mul, d, st()))
                else:
                    wd = weight_decay if opt_name == "adamw" else 0.0
                    check(lib.m355_adamw_step(flat_p.data_ptr(), flat_g.data_ptr(), state1.data_ptr(), state2.data_ptr(),
                                              ema.data_ptr(), eng.group.data_ptr(), n_train, lr, lr_bias, mom, 0.999, 1e-8,
                                              wd, opt_steps, grad_mul, d, st()))
                ema[n_train:].mul_(d).add_(flat_p[n_train:], alpha=1.0 - d)          # BN running statistics
                eng.repack()
            else:
                skipped += 1
            scaler.update(found_inf)
            global_step += 1
            if max_steps and global_step >= max_steps:
                stop = True
                break
        mloss = (mloss / max(i + 1, 1)).tolist()
        if world > 1:
            dist.barrier()
        if rank == 0:
            sd = eng.state_dict(flat=ema)
            if validator is not None:
                last_res = validator(sd)
            row = {"epoch": epoch + 1, "time": round(time.time() - t_start, 3), "train/box_loss": mloss[0], "train/seg_loss": mloss[1],
                   "train/cls_loss": mloss[2], "train/dfl_loss": mloss[3], "lr/pg0": lr, "loss_scale": scaler.scale, **last_res}
            rows.append(row)
            if a.save:
                with open(os.path.join(save_dir, "results.csv"), "w", newline="") as fcsv:
                    w = csv.DictWriter(fcsv, fieldnames=fields, extrasaction="ignore")
                    w.writeheader()
                    for rr in rows:
                        w.writerow({k: (f"{v:.5g}" if isinstance(v, float) else v) for k, v in rr.items()})
                fit = last_res.get("fitness", -mloss[0] - mloss[1] - mloss[2] - mloss[3])
                improved = fit > best_fit or not os.path.isfile(os.path.join(wdir, "best.pt"))
                if improved:
                    best_fit, best_epoch = max(fit, best_fit), epoch
                    ckpt(os.path.join(wdir, "best.pt"), sd, epoch + 1)
                ckpt(os.path.join(wdir, "last.pt"), sd, epoch + 1, with_trainer=True)
            if a.patience and best_epoch >= 0 and epoch - best_epoch >= a.patience:     # early stopping (A14: patience 100)
                stop = True
                if a.verbose:
                    print(f"early stop: no fitness improvement for {a.patience} epochs (best epoch {best_epoch + 1})")
            if a.verbose:
                msg = (f"epoch {epoch + 1}/{epochs}  box {mloss[0]:.4f} seg {mloss[1]:.4f} cls {mloss[2]:.4f} dfl {mloss[3]:.4f}"
                       f"  lr {lr:.2e} scale {scaler.scale:g}")
                if last_res:
                    msg += f"  mAP50(B) {last_res['metrics/mAP50(B)']:.4f} mAP50(M) {last_res['metrics/mAP50(M)']:.4f}"
                print(msg, flush=True)
        if world > 1:
            flag = torch.tensor([1 if stop else 0], device=dev)
            dist.broadcast(flag, src=0)                # rank 0 decides (it owns the validator)
            stop = bool(int(flag.item()))
        if stop:
            break
    if validator is not None:
        validator.close()
    final_sd = eng.state_dict(flat=ema)
    if world > 1 and own_pg:
        dist.destroy_process_group()
    best = os.path.join(wdir, "best.pt")
    if rank == 0 and a.save and os.path.isfile(best):
        model._load_checkpoint(best)
    else:
        model.state_dict = final_sd
    model._drop_engines()
    out = _metrics_namespace(last_res, save_dir) if last_res else SimpleNamespace(save_dir=save_dir, results_dict={})
    out.history, out.skipped_steps, out.optimizer_steps = rows, skipped, opt_steps
    if rank == 0 and a.verbose:
        print(f"{epochs} epochs completed in {(time.time() - t_start) / 3600:.3f} hours.  Results saved to {save_dir}")
    return out
