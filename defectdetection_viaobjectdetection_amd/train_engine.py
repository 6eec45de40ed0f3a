"""Train-mode forward / backward of YOLOv8-seg on the HIP kernels (SURVEY.md A13, A16 building blocks).

Stands where ``SegmentationModel.forward`` in training mode + PyTorch autograd stand upstream (call site
/root/reference/BscanBased/yolo_seg_train.py:12-19).  The graph, buffers and the order of kernel launches are
orchestrated here in Python over the raw C-ABI launches of ``libmi355yolo.so``:

  forward  per Conv block : implicit-GEMM / halo conv (no bias, no act) -> z ; train-mode BatchNorm (batch
                            statistics) + SiLU (+ residual) -> activation slice      [conv_launch, bn_train_fwd]
  backward per Conv block : BN+SiLU backward -> dz, dgamma, dbeta ; wgrad (pixel-axis GEMM) -> dW ; dgrad
                            (same conv kernels, flipped / transposed-stride weights) accumulated into the
                            producer's gradient slice                                [bn_train_bwd, wgrad, conv]

Every FLOP-carrying op runs in the hand-written kernels.  PyTorch is used for buffers, for re-packing the fp32
master weights to fp16 GEMM layouts after an optimizer step, and for the byte-moving leftovers of the backward
pass that are <1 % of the traffic (residual / concat gradient adds, 2x2 upsample-sum, SPPF max-pool routing,
bias-gradient sums).  Master parameters are fp32 in KRSC layout = the layout wgrad produces.
"""
from __future__ import annotations

import ctypes as C
import os
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from ._capi import ConvLaunchArgs, WgradLaunchArgs, check, lib
from .spec import BN_EPS, NM, REG_MAX, SCALES, V9C, ConvSpec, _make_divisible, conv_branches, conv_specs

BN_MOMENTUM = 0.03


def _ceil(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class Slice:
    t: int
    off: int
    c: int


class TrainEngine:
    def __init__(self, scale: str = "n", nc: int = 1, imgsz: Tuple[int, int] = (640, 640), batch: int = 2,
                 device: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("TrainEngine needs a gfx950 GPU; there is no CPU fallback")
        self.dev = torch.device("cuda", device)
        self.scale, self.nc, self.imgsz, self.B = scale, nc, tuple(imgsz), batch
        # RepConvN (yolov9c-seg) trains in its un-merged form: the 3x3 and the 1x1 branch are two Conv + BN of their own
        # (state-dict prefixes <name>.conv1 / <name>.conv2), summed before the SiLU
        self.specs: Dict[str, ConvSpec] = {}
        for s in conv_specs(scale, nc):
            if s.rep:
                for pre, k in conv_branches(s):
                    self.specs[pre] = ConvSpec(pre, s.cin, s.cout, k, 1, True)
            else:
                self.specs[s.name] = s
        self.galias: Dict[int, int] = {}     # tensor -> tensor whose gradient buffer it shares (the two RepConvN branches)
        self.tensors: List[torch.Tensor] = []
        self.gtensors: List[Optional[torch.Tensor]] = []
        self.ops: List[dict] = []
        self.params: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        self.packed: Dict[str, torch.Tensor] = {}
        self.saved: Dict[str, dict] = {}
        self.zero_page = torch.zeros(256, dtype=torch.uint8, device=self.dev)
        self.zero_bias = torch.zeros(4096, dtype=torch.float32, device=self.dev)
        self.wgrad_ws = torch.empty(1 << 20, dtype=torch.float32, device=self.dev)   # split-K partial slabs (deterministic wgrad)
        self._colsum_ws: Dict[int, torch.Tensor] = {}                                # partial rows of the bias-gradient column sums
        # weight gradients run on a side stream beside the input gradient of the same layer (both only read dZ): on the small
        # maps neither kernel fills 256 CUs on its own.  M355_NO_WGRAD_STREAM=1: everything on the caller's stream.
        # Side streams only in a single-process job.  With two ranks sharing one GPU (the gloo rehearsal of the N > 1 path) every
        # cross-stream dependency waited for a queue time slice: 13 s per step instead of 0.16 s; on one GPU per rank that
        # should not happen, but it cannot be verified here, so a multi-rank job keeps the one-stream order it was tested with
        # (M355_SIDE_STREAMS=1 forces them on).
        import torch.distributed as _dist
        multi = _dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1
        if multi and os.environ.get("M355_SIDE_STREAMS") != "1":
            os_no_side = True
        else:
            os_no_side = False
        self._wg_stream = None if (os_no_side or os.environ.get("M355_NO_WGRAD_STREAM") == "1") else torch.cuda.Stream(device=self.dev)
        self._wg_done = None                                                         # event after the last side-stream launch
        # input gradients of the stride-2 3x3 convs as four phase convs over dY (16 tap slots for 9 taps) instead of the masked
        # transposed-stride gather (36): M355_NO_DGRAD_PHASES=1 restores the gather (A/B and the parity test)
        self._dgrad_phases = os.environ.get("M355_NO_DGRAD_PHASES") != "1"
        # forward: the 1/8-level head + the prototype branch beside the rest of the neck (M355_NO_HEAD_STREAM=1: one stream)
        self._head_stream = None if (os_no_side or os.environ.get("M355_NO_HEAD_STREAM") == "1") else torch.cuda.Stream(device=self.dev)
        self._head_ops = None
        self._build()

    # ------------------------------------------------------------------ graph
    def _tensor(self, H, W, Cc) -> int:
        self.tensors.append(torch.zeros((self.B, H, W, Cc), dtype=torch.float16, device=self.dev))
        self.gtensors.append(None)
        return len(self.tensors) - 1

    def _conv(self, name, src: Slice, dst: Slice, res: Optional[Slice] = None, act: int = 1):
        s = self.specs[name]
        assert s.cin == src.c or (s.cin == 3 and src.c == 8), (name, s.cin, src.c)
        assert s.cout == dst.c, (name, s.cout, dst.c)
        self.ops.append(dict(kind="conv", name=name, src=src, dst=dst, res=res, k=s.k, s=s.stride, act=act))

    def _c2f(self, name, src: Slice, dst: Slice, n: int, shortcut: bool):
        H, W = self.tensors[src.t].shape[1:3]
        c = dst.c // 2
        cat = self._tensor(H, W, (2 + n) * c)
        self._conv(f"{name}.cv1", src, Slice(cat, 0, 2 * c))
        for j in range(n):
            tmp = self._tensor(H, W, c)
            s_in = Slice(cat, (1 + j) * c, c)
            self._conv(f"{name}.m.{j}.cv1", s_in, Slice(tmp, 0, c))
            self._conv(f"{name}.m.{j}.cv2", Slice(tmp, 0, c), Slice(cat, (2 + j) * c, c), s_in if shortcut else None)
        self._conv(f"{name}.cv2", Slice(cat, 0, (2 + n) * c), dst)

    # ---- yolov9c-seg blocks (oracle/yolov9c_seg_oracle.py; /root/reference/BscanBased/yolo_seg_train.py:7 builds this graph)
    def _repconv(self, name, src: Slice, dst: Slice):
        """RepConvN: dst = SiLU(BN(conv3x3(src)) + BN(conv1x1(src))).  The two branches run as activation-free Conv + BN ops
        into temporaries that share ONE gradient buffer (both receive d SiLU(a + b) / d(a + b) x the incoming gradient)."""
        H, W = self.tensors[src.t].shape[1:3]
        ta, tb = self._tensor(H, W, dst.c), self._tensor(H, W, dst.c)
        self.galias[tb] = ta
        self._conv(f"{name}.conv1", src, Slice(ta, 0, dst.c), act=0)
        self._conv(f"{name}.conv2", src, Slice(tb, 0, dst.c), act=0)
        self.ops.append(dict(kind="addsilu", name=name, a=Slice(ta, 0, dst.c), b=Slice(tb, 0, dst.c), dst=dst))

    def _repcsp(self, name, src: Slice, dst: Slice):
        H, W = self.tensors[src.t].shape[1:3]
        c_ = dst.c // 2
        icat, u, r = self._tensor(H, W, 2 * c_), self._tensor(H, W, c_), self._tensor(H, W, c_)
        self._conv(f"{name}.cv1", src, Slice(u, 0, c_))
        self._repconv(f"{name}.m.0.cv1", Slice(u, 0, c_), Slice(r, 0, c_))                       # RepBottleneck: x + cv2(cv1(x))
        self._conv(f"{name}.m.0.cv2", Slice(r, 0, c_), Slice(icat, 0, c_), Slice(u, 0, c_))
        self._conv(f"{name}.cv2", src, Slice(icat, c_, c_))
        self._conv(f"{name}.cv3", Slice(icat, 0, 2 * c_), dst)

    def _elan(self, name, src: Slice, dst: Slice, c3: int, c4: int):
        """RepNCSPELAN4: y = chunk2(cv1(x)); y += [cv2(y[-1])]; y += [cv3(y[-1])]; cv4(cat(y)); zero-copy concat."""
        H, W = self.tensors[src.t].shape[1:3]
        cat = self._tensor(H, W, c3 + 2 * c4)
        ta, tb = self._tensor(H, W, c4), self._tensor(H, W, c4)
        self._conv(f"{name}.cv1", src, Slice(cat, 0, c3))
        self._repcsp(f"{name}.cv2.0", Slice(cat, c3 // 2, c3 // 2), Slice(ta, 0, c4))
        self._conv(f"{name}.cv2.1", Slice(ta, 0, c4), Slice(cat, c3, c4))
        self._repcsp(f"{name}.cv3.0", Slice(cat, c3, c4), Slice(tb, 0, c4))
        self._conv(f"{name}.cv3.1", Slice(tb, 0, c4), Slice(cat, c3 + c4, c4))
        self._conv(f"{name}.cv4", Slice(cat, 0, c3 + 2 * c4), dst)

    def _adown(self, name, src: Slice, dst: Slice):
        """ADown: x = avg_pool2d(x, 2, 1, 0); x1, x2 = chunk2(x); cat(Conv3x3/s2(x1), Conv1x1(max_pool2d(x2, 3, 2, 1)))."""
        H, W = self.tensors[src.t].shape[1:3]
        ch = src.c // 2
        p1, p2 = self._tensor(H - 1, W - 1, ch), self._tensor(H // 2, W // 2, ch)
        self.ops.append(dict(kind="adown", src=src, p1=Slice(p1, 0, ch), p2=Slice(p2, 0, ch)))
        self._conv(f"{name}.cv1", Slice(p1, 0, ch), Slice(dst.t, dst.off, dst.c // 2))
        self._conv(f"{name}.cv2", Slice(p2, 0, ch), Slice(dst.t, dst.off + dst.c // 2, dst.c // 2))

    def _build_v9c(self):
        H, W = self.imgsz
        T = self._tensor
        self.x8 = T(H, W, 8)
        cat11 = T(H // 16, W // 16, 1024)          # [up(x9), x6]
        cat14 = T(H // 8, W // 8, 1024)            # [up(x12), x4]
        cat17 = T(H // 16, W // 16, 768)           # [adown(x15), x12]
        cat20 = T(H // 32, W // 32, 1024)          # [adown(x18), x9]
        x4, x6 = Slice(cat14, 512, 512), Slice(cat11, 512, 512)
        x9, x12 = Slice(cat20, 512, 512), Slice(cat17, 256, 512)
        t0, t1, t2 = T(H // 2, W // 2, 64), T(H // 4, W // 4, 128), T(H // 4, W // 4, 256)
        self._conv("model.0", Slice(self.x8, 0, 8), Slice(t0, 0, 64))
        self._conv("model.1", Slice(t0, 0, 64), Slice(t1, 0, 128))
        self._elan("model.2", Slice(t1, 0, 128), Slice(t2, 0, 256), 128, 64)
        t3 = T(H // 8, W // 8, 256)
        self._adown("model.3", Slice(t2, 0, 256), Slice(t3, 0, 256))
        self._elan("model.4", Slice(t3, 0, 256), x4, 256, 128)
        t5 = T(H // 16, W // 16, 512)
        self._adown("model.5", x4, Slice(t5, 0, 512))
        self._elan("model.6", Slice(t5, 0, 512), x6, 512, 256)
        t7, t8 = T(H // 32, W // 32, 512), T(H // 32, W // 32, 512)
        self._adown("model.7", x6, Slice(t7, 0, 512))
        self._elan("model.8", Slice(t7, 0, 512), Slice(t8, 0, 512), 512, 256)
        sp = T(H // 32, W // 32, 1024)             # SPPELAN = SPPF's shape: cv1, three serial 5x5 max-pools, cv5 over the concat
        self._conv("model.9.cv1", Slice(t8, 0, 512), Slice(sp, 0, 256))
        self.ops.append(dict(kind="pool", src=Slice(sp, 0, 256), dst=Slice(sp, 256, 768)))
        self._conv("model.9.cv5", Slice(sp, 0, 1024), x9)
        self.ops.append(dict(kind="up", src=x9, dst=Slice(cat11, 0, 512)))
        self._elan("model.12", Slice(cat11, 0, 1024), x12, 512, 256)
        self.ops.append(dict(kind="up", src=x12, dst=Slice(cat14, 0, 512)))
        t15 = T(H // 8, W // 8, 256)
        self._elan("model.15", Slice(cat14, 0, 1024), Slice(t15, 0, 256), 256, 128)
        self._adown("model.16", Slice(t15, 0, 256), Slice(cat17, 0, 256))
        t18 = T(H // 16, W // 16, 512)
        self._elan("model.18", Slice(cat17, 0, 768), Slice(t18, 0, 512), 512, 256)
        self._adown("model.19", Slice(t18, 0, 512), Slice(cat20, 0, 512))
        t21 = T(H // 32, W // 32, 512)
        self._elan("model.21", Slice(cat20, 0, 1024), Slice(t21, 0, 512), 512, 256)
        self._build_head((t15, t18, t21), (256, 512, 512), 256)

    def _build(self):
        if self.scale == V9C:
            return self._build_v9c()
        depth, width, maxc = SCALES[self.scale]
        ch = lambda c: _make_divisible(min(c, maxc) * width, 8)  # noqa: E731
        rep = lambda n: max(round(n * depth), 1) if n > 1 else n  # noqa: E731
        c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
        H, W = self.imgsz
        T = self._tensor
        self.x8 = T(H, W, 8)  # input: 3 channels / 255 in fp16, zero-padded to 8 channels
        cat11 = T(H // 16, W // 16, c1024 + c512)
        cat14 = T(H // 8, W // 8, c512 + c256)
        cat17 = T(H // 16, W // 16, c256 + c512)
        cat20 = T(H // 32, W // 32, c512 + c1024)
        x4, x6 = Slice(cat14, c512, c256), Slice(cat11, c1024, c512)
        x9, x12 = Slice(cat20, c512, c1024), Slice(cat17, c256, c512)
        t0 = T(H // 2, W // 2, c64)
        self._conv("model.0", Slice(self.x8, 0, 8), Slice(t0, 0, c64))
        t1 = T(H // 4, W // 4, c128)
        self._conv("model.1", Slice(t0, 0, c64), Slice(t1, 0, c128))
        t2 = T(H // 4, W // 4, c128)
        self._c2f("model.2", Slice(t1, 0, c128), Slice(t2, 0, c128), rep(3), True)
        t3 = T(H // 8, W // 8, c256)
        self._conv("model.3", Slice(t2, 0, c128), Slice(t3, 0, c256))
        self._c2f("model.4", Slice(t3, 0, c256), x4, rep(6), True)
        t5 = T(H // 16, W // 16, c512)
        self._conv("model.5", x4, Slice(t5, 0, c512))
        self._c2f("model.6", Slice(t5, 0, c512), x6, rep(6), True)
        t7 = T(H // 32, W // 32, c1024)
        self._conv("model.7", x6, Slice(t7, 0, c1024))
        t8 = T(H // 32, W // 32, c1024)
        self._c2f("model.8", Slice(t7, 0, c1024), Slice(t8, 0, c1024), rep(3), True)
        c_ = c1024 // 2
        sp = T(H // 32, W // 32, 4 * c_)
        self._conv("model.9.cv1", Slice(t8, 0, c1024), Slice(sp, 0, c_))
        self.ops.append(dict(kind="pool", src=Slice(sp, 0, c_), dst=Slice(sp, c_, 3 * c_)))
        self._conv("model.9.cv2", Slice(sp, 0, 4 * c_), x9)
        self.ops.append(dict(kind="up", src=x9, dst=Slice(cat11, 0, c1024)))
        self._c2f("model.12", Slice(cat11, 0, c1024 + c512), x12, rep(3), False)
        self.ops.append(dict(kind="up", src=x12, dst=Slice(cat14, 0, c512)))
        t15 = T(H // 8, W // 8, c256)
        self._c2f("model.15", Slice(cat14, 0, c512 + c256), Slice(t15, 0, c256), rep(3), False)
        self._conv("model.16", Slice(t15, 0, c256), Slice(cat17, 0, c256))
        t18 = T(H // 16, W // 16, c512)
        self._c2f("model.18", Slice(cat17, 0, c256 + c512), Slice(t18, 0, c512), rep(3), False)
        self._conv("model.19", Slice(t18, 0, c512), Slice(cat20, 0, c512))
        t21 = T(H // 32, W // 32, c1024)
        self._c2f("model.21", Slice(cat20, 0, c512 + c1024), Slice(t21, 0, c1024), rep(3), False)
        self._build_head((t15, t18, t21), (c256, c512, c1024), ch(256))

    def _build_head(self, feats, fch, npr):
        """model.22 = Segment on the three feature tensors (shared by the yolov8-seg and the yolov9c-seg graphs)."""
        H, W = self.imgsz
        T = self._tensor
        t15, c256 = feats[0], fch[0]
        hc2, hc3, hc4 = max(16, fch[0] // 4, REG_MAX * 4), max(fch[0], min(self.nc, 100)), max(fch[0] // 4, NM)
        hw = [(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
        self.level_n = [h * w for h, w in hw]
        self.A = sum(self.level_n)
        self.rw = 64 + self.nc + NM
        self.raw = torch.zeros((self.B, self.A, self.rw), dtype=torch.float32, device=self.dev)
        off = 0
        for l in range(3):
            f = Slice(feats[l], 0, fch[l])
            for br, hc, cout, choff in (("cv2", hc2, 64, 0), ("cv3", hc3, self.nc, 64), ("cv4", hc4, NM, 64 + self.nc)):
                u1, u2 = T(*hw[l], hc), T(*hw[l], hc)
                self._conv(f"model.22.{br}.{l}.0", f, Slice(u1, 0, hc))
                self._conv(f"model.22.{br}.{l}.1", Slice(u1, 0, hc), Slice(u2, 0, hc))
                self.ops.append(dict(kind="plain", name=f"model.22.{br}.{l}.2", src=Slice(u2, 0, hc), cout=cout,
                                     level_off=off, ch_off=choff, hw=hw[l]))
            off += self.level_n[l]
        pr1, pr2, pr3 = T(H // 8, W // 8, npr), T(H // 4, W // 4, npr), T(H // 4, W // 4, npr)
        self.protos_t = T(H // 4, W // 4, NM)
        self._conv("model.22.proto.cv1", Slice(t15, 0, c256), Slice(pr1, 0, npr))
        self.ops.append(dict(kind="convt", name="model.22.proto.upsample", src=Slice(pr1, 0, npr), dst=Slice(pr2, 0, npr)))
        self._conv("model.22.proto.cv2", Slice(pr2, 0, npr), Slice(pr3, 0, npr))
        self._conv("model.22.proto.cv3", Slice(pr3, 0, npr), Slice(self.protos_t, 0, NM))

    # ------------------------------------------------------------------ parameters
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Upstream-named state dict (OIHW conv weights) -> fp32 master parameters (KRSC) on the device."""
        for s in self.specs.values():
            if s.has_bn:
                w = sd[f"{s.name}.conv.weight"].float().permute(0, 2, 3, 1).contiguous()
                self.params[f"{s.name}.conv.weight"] = w.to(self.dev)
                for p in ("bias", "weight", "running_mean", "running_var"):      # bias, weight adjacent: = BN-bwd output
                    self.params[f"{s.name}.bn.{p}"] = sd[f"{s.name}.bn.{p}"].float().clone().to(self.dev)
            elif s.transposed:
                self.params[f"{s.name}.weight"] = sd[f"{s.name}.weight"].float().clone().to(self.dev)  # (cin,cout,2,2)
                self.params[f"{s.name}.bias"] = sd[f"{s.name}.bias"].float().clone().to(self.dev)
            else:
                self.params[f"{s.name}.weight"] = sd[f"{s.name}.weight"].float().permute(0, 2, 3, 1).contiguous().to(self.dev)
                self.params[f"{s.name}.bias"] = sd[f"{s.name}.bias"].float().clone().to(self.dev)
        self._flatten()
        self.repack()

    def _flatten(self) -> None:
        """Re-home parameters, buffers and gradients in flat fp32 buffers (views keep their names and shapes):
        ``flat_params[:n_train]`` trainable, the tail = BN running statistics; ``flat_grads`` matches the head.
        One buffer = one optimizer launch and one all-reduce.  ``group`` marks 0 weights / 1 norm weights / 2 biases."""
        train = [k for k in self.params if "running_" not in k]
        bufs = [k for k in self.params if "running_" in k]
        n_train = sum(self.params[k].numel() for k in train)
        n_all = n_train + sum(self.params[k].numel() for k in bufs)
        flat = torch.empty(n_all, device=self.dev)
        self.flat_grads = torch.zeros(n_train, device=self.dev)
        self.group = torch.empty(n_train, dtype=torch.uint8, device=self.dev)
        o = 0
        self.layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        for k in train + bufs:
            v = self.params[k]
            n = v.numel()
            self.layout[k] = (o, tuple(v.shape))
            flat[o:o + n].copy_(v.reshape(-1))
            self.params[k] = flat[o:o + n].view(v.shape)
            if o < n_train:
                self.grads[k] = self.flat_grads[o:o + n].view(v.shape)
                self.group[o:o + n] = 2 if k.endswith(".bias") else (1 if ".bn." in k else 0)
            o += n
        self.flat_params, self.n_train = flat, n_train

    def grad_spans(self) -> Dict[str, Tuple[int, int]]:
        """{name: (offset, numel)} of the trainable parameters inside ``flat_grads`` (for GradBucketReducer)."""
        return {k: (o, int(torch.Size(sh).numel())) for k, (o, sh) in self.layout.items() if o < self.n_train}

    def state_dict(self, flat: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Upstream-named state dict (OIHW weights, CPU).  ``flat``: read the values from another flat buffer of the
        same layout (the EMA copy) instead of the live parameters."""
        out = {}
        src = self.params if flat is None else {k: flat[o:o + torch.Size(sh).numel()].view(sh)
                                                for k, (o, sh) in self.layout.items()}
        for k, v in src.items():
            name = k.rsplit(".", 1)[0].replace(".conv", "")
            s = self.specs.get(name) or self.specs.get(k.rsplit(".", 2)[0])
            if k.endswith("weight") and v.dim() == 4 and not (s and s.transposed):
                out[k] = v.permute(0, 3, 1, 2).contiguous().cpu()
            else:
                out[k] = v.clone().cpu()
        for s in self.specs.values():
            if s.has_bn:
                out[f"{s.name}.bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        out["model.22.dfl.conv.weight"] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
        return out

    def trainable(self) -> List[Tuple[str, torch.Tensor, torch.Tensor]]:
        return [(k, self.params[k], self.grads[k]) for k in self.grads]

    def _pack_view(self, key: str, rows: int, K: int, shape) -> torch.Tensor:
        """View of shape `shape` (rows x K elements, row-major) onto the top-left corner of the fp16
        [rows padded to 128][K padded to 64] GEMM matrix of `key` (allocated and zeroed on first use).  A re-pack is
        then ONE strided convert-copy per layout into this view instead of ~8 small launches and an allocation."""
        out = self.packed.get(key)
        if out is None:
            out = torch.zeros((_ceil(rows, 128), _ceil(K, 64)), dtype=torch.float16, device=self.dev)
            self.packed[key] = out
        return out[:rows, :K].view(shape)

    def _repack_pairs(self):
        """(destination fp16 view, source fp32 view, flipped dims) of every strided convert-copy of a re-pack: forward
        [Cout][(kh,kw,ci)] and dgrad [Cin][(kh',kw',co)] layouts.  Padded rows / channels (stem 3 -> 8 input channels,
        nc -> 8 class rows) stay zero from the allocation."""
        out = []
        for s in self.specs.values():
            if s.transposed:
                w = self.params[f"{s.name}.weight"]                       # (cin, cout, 2, 2)
                out.append((self._pack_view(s.name + ":fwd", 4 * s.cout, s.cin, (2, 2, s.cout, s.cin)), w.permute(2, 3, 1, 0), ()))
                out.append((self._pack_view(s.name + ":dgrad", s.cin, 4 * s.cout, (s.cin, 2, 2, s.cout)), w.permute(0, 2, 3, 1), ()))
                continue
            w = self.params[f"{s.name}.conv.weight" if s.has_bn else f"{s.name}.weight"]  # (cout, k, k, cin)
            k = s.k
            cin_p = 8 if s.cin == 3 else s.cin                              # stem: input padded to 8 channels
            cout_p = _ceil(s.cout, 8)                                       # class branch (nc) -> rows padded to 8
            out.append((self._pack_view(s.name + ":fwd", cout_p, k * k * cin_p, (cout_p, k, k, cin_p))[:s.cout, :, :, :s.cin], w, ()))
            if s.cin != 3:                                                  # stride 2 = transposed-stride gather, no flip
                out.append((self._pack_view(s.name + ":dgrad", cin_p, k * k * cout_p, (cin_p, k, k, cout_p))[..., :s.cout],
                            w.permute(3, 1, 2, 0), () if s.stride == 2 else (1, 2)))
                if s.stride == 2 and k == 3 and self._dgrad_phases:
                    # the same gradient as four 2x2 phase convs over dY (m355_conv_launch, tmode 2): rows [phase][ci], columns
                    # [(ty, tx)][co]; dX row 2i (+0) takes tap kh = 1 from dY row i, row 2i + 1 takes kh = 2 from row i and kh = 0
                    # from row i + 1 (columns alike); the seven unused tap slots of the sixteen stay zero from the allocation
                    v = self._pack_view(s.name + ":dgrad4", 4 * cin_p, 4 * cout_p, (4, cin_p, 2, 2, cout_p))
                    wt = w.permute(3, 1, 2, 0)                               # (cin, kh, kw, cout)
                    taps = {0: ((0, 1),), 1: ((0, 2), (1, 0))}              # parity -> ((window slot, forward tap), ...)
                    # forward input channels in multiples of 64: a phase's taps COMPACT on its K axis (slot ty * (1 + b) + tx), its K loop
                    # ends behind them (9 tap slots over the four phases instead of 16); otherwise window slots ty * 2 + tx
                    vf = v.view(4, cin_p, 4, cout_p)
                    for a in (0, 1):
                        for b in (0, 1):
                            for ty, kh in taps[a]:
                                for tx, kw in taps[b]:
                                    slot = ty * (1 + b) + tx if cin_p % 64 == 0 else ty * 2 + tx
                                    out.append((vf[2 * a + b, :, slot:slot + 1, :s.cout].unsqueeze(2), wt[:, kh:kh + 1, kw:kw + 1, :], ()))
        return out

    def _repack_torch(self) -> None:
        """The re-pack as one torch convert-copy per layout (the form the job kernel replaced; kept for its test)."""
        for dst, src, flips in self._repack_pairs():
            dst.copy_(src.flip(*flips) if flips else src)

    def repack(self) -> None:
        """fp32 master weights -> fp16 GEMM layouts in ONE launch (m355_repack_launch): the job table is built once, the
        parameter and layout buffers of an engine never move."""
        if getattr(self, "_repack_jobs", None) is None:
            import numpy as np
            dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("n", "<i4", 4), ("ss", "<i8", 4), ("ds", "<i8", 4), ("block0", "<i4"), ("pad", "<i4")])
            jobs, block_job = [], []
            for dst, src, flips in self._repack_pairs():
                assert tuple(dst.shape) == tuple(src.shape) and dst.dim() == 4 and dst.dtype == torch.float16 and src.dtype == torch.float32
                n, ss, ds = list(dst.shape), list(src.stride()), list(dst.stride())
                sp = src.data_ptr()
                for d in flips:                                             # a flip = start at the far end, negative stride
                    sp += (n[d] - 1) * ss[d] * 4
                    ss[d] = -ss[d]
                total = n[0] * n[1] * n[2] * n[3]
                if total == 0:
                    continue
                jobs.append((sp, dst.data_ptr(), n, ss, ds, len(block_job), 0))
                block_job += [len(jobs) - 1] * ((total + 1023) // 1024)
            arr = np.array(jobs, dtype=dt)
            self._repack_jobs = torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(self.dev)
            self._repack_blocks = torch.tensor(block_job, dtype=torch.int32, device=self.dev)
        check(lib.m355_repack_launch(C.c_void_p(self._repack_jobs.data_ptr()), C.c_void_p(self._repack_blocks.data_ptr()),
                                     int(self._repack_blocks.numel()), self._stream()))

    # ------------------------------------------------------------------ kernel plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @contextmanager
    def _beside(self, op):
        """Run the enclosed launches (a layer's weight gradient and the copies behind it) on the side stream, after everything
        enqueued so far on the caller's stream; `_join_side()` makes the caller's stream wait for them."""
        if self._wg_stream is None:
            yield
            return
        ev = op.get("_ev_fork")
        if ev is None:
            ev = op["_ev_fork"] = torch.cuda.Event()
            op["_ev_join"] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._wg_stream.wait_event(ev)
        with torch.cuda.stream(self._wg_stream):
            yield
            op["_ev_join"].record(self._wg_stream)
        self._wg_done = op["_ev_join"]

    def _join_side(self):
        if self._wg_done is not None:
            torch.cuda.current_stream().wait_event(self._wg_done)
            self._wg_done = None

    def _padded_bias(self, op, name: str) -> torch.Tensor:
        """The bias of a plain conv followed by 128 zeros (the kernels read whole channel tiles): a persistent buffer refreshed
        with one copy per step."""
        b = self.params[f"{name}.bias"]
        buf = op.get("_bias_pad")
        if buf is None:
            buf = op["_bias_pad"] = torch.zeros(b.numel() + 128, dtype=torch.float32, device=self.dev)
        buf[:b.numel()].copy_(b)
        return buf

    def _colsum(self, src_ptr: int, f16: bool, nb: int, bstride: int, rows: int, ld: int, cols: int, out: torch.Tensor) -> torch.Tensor:
        """out[c] = sum over (b, r) of src[b * bstride + r * ld + c] in a fixed order (bias gradients); out fp32 contiguous."""
        need = int(lib.m355_colsum_workspace_floats(nb, cols))
        key = torch.cuda.current_stream().cuda_stream                       # one workspace per stream that runs column sums
        ws = self._colsum_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._colsum_ws[key] = torch.empty(need, dtype=torch.float32, device=self.dev)
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == cols
        check(lib.m355_colsum_launch(src_ptr, 1 if f16 else 0, nb, bstride, rows, ld, cols, ws.data_ptr(), out.data_ptr(),
                                     self._stream()))
        return out

    def _slice_ptr(self, tensors, sl: Slice):
        t = tensors[sl.t]
        return t.data_ptr() + sl.off * t.element_size(), t.shape[1] * t.shape[2] * t.shape[3], t.shape[3]

    def _conv_launch(self, x_ptr, x_bs, ldx, hi, wi, cin, w, y_ptr, y_bs, ldy, ho, wo, cout, k, stride, pad, bias=None,
                     res_ptr=0, r_bs=0, ldr=0, act=0, out_f32=0, convt_co=0, tmode=0):
        a = ConvLaunchArgs()
        a.x, a.x_bstride, a.ldx, a.hi, a.wi, a.cin = x_ptr, x_bs, ldx, hi, wi, cin
        a.w_packed, a.kpad = w.data_ptr(), w.shape[1]
        a.bias = (bias if bias is not None else self.zero_bias).data_ptr()
        a.y, a.y_bstride, a.ldy, a.ho, a.wo, a.cout = y_ptr, y_bs, ldy, ho, wo, cout
        a.res, a.r_bstride, a.ldr = res_ptr, r_bs, ldr
        a.ksize, a.stride, a.pad, a.batch = k, stride, pad, self.B
        a.act, a.out_f32, a.convt_co, a.tmode = act, out_f32, convt_co, tmode
        a.zero_page = self.zero_page.data_ptr()
        check(lib.m355_conv_launch(C.byref(a), self._stream()))

    def _wgrad_launch(self, dz, lddz, dz_bs, x_ptr, x_bs, ldx, hi, wi, cin, ho, wo, cout, k, stride, pad, dw):
        a = WgradLaunchArgs()
        a.dz, a.dz_bstride, a.lddz = dz, dz_bs, lddz
        a.x, a.x_bstride, a.ldx = x_ptr, x_bs, ldx
        a.hi, a.wi, a.cin, a.ho, a.wo, a.cout = hi, wi, cin, ho, wo, cout
        a.ksize, a.stride, a.pad, a.batch = k, stride, pad, self.B
        a.dw, a.zero_page = dw.data_ptr(), self.zero_page.data_ptr()
        need = int(lib.m355_wgrad_workspace_bytes(self.B, ho, wo, cin, cout, k))
        if need > self.wgrad_ws.numel() * 4:                          # grows to the largest layer, then stays
            torch.cuda.current_stream().synchronize()                  # (launches in flight still use the old workspace)
            self.wgrad_ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.dev)
        a.ws, a.ws_bytes = self.wgrad_ws.data_ptr(), self.wgrad_ws.numel() * 4
        check(lib.m355_wgrad_launch(C.byref(a), self._stream()))

    def _fwd_op(self, op, update_running_stats: bool) -> None:
        """Enqueue one forward op on the current stream."""
        B = self.B
        st = self._stream()
        kind = op["kind"]
        if kind == "conv":
            name, src, dst = op["name"], op["src"], op["dst"]
            tin, tout = self.tensors[src.t], self.tensors[dst.t]
            hi, wi = tin.shape[1:3]
            ho, wo = tout.shape[1:3]
            cout = dst.c
            sv = self.saved.setdefault(name, {})
            if "z" not in sv:
                sv["z"] = torch.empty((B, ho, wo, cout), dtype=torch.float16, device=self.dev)
                sv["mean"] = torch.empty(cout, device=self.dev)
                sv["invstd"] = torch.empty(cout, device=self.dev)
                # per-block partial sums + ticket of the ordered batch-norm reductions: zeroed once, self-resetting
                sv["ws"] = torch.zeros(int(lib.m355_bn_workspace_floats(cout)), device=self.dev)
            xp, xbs, ldx = self._slice_ptr(self.tensors, src)
            self._conv_launch(xp, xbs, ldx, hi, wi, src.c, self.packed[name + ":fwd"], sv["z"].data_ptr(), ho * wo * cout,
                              cout, ho, wo, cout, op["k"], op["s"], op["k"] // 2)
            yp, _, ldy = self._slice_ptr(self.tensors, dst)
            rp, ldr = (0, 0)
            if op["res"] is not None:
                rp, _, ldr = self._slice_ptr(self.tensors, op["res"])
            rm = self.params[f"{name}.bn.running_mean"].data_ptr() if update_running_stats else 0
            rv = self.params[f"{name}.bn.running_var"].data_ptr() if update_running_stats else 0
            check(lib.m355_bn_train_fwd_launch(sv["z"].data_ptr(), B * ho * wo, cout, cout,
                                               self.params[f"{name}.bn.weight"].data_ptr(),
                                               self.params[f"{name}.bn.bias"].data_ptr(), BN_EPS, op.get("act", 1), yp, ldy, rp, ldr,
                                               sv["mean"].data_ptr(), sv["invstd"].data_ptr(), sv["ws"].data_ptr(),
                                               rm, rv, BN_MOMENTUM, st))
        elif kind == "plain":
            name, src = op["name"], op["src"]
            tin = self.tensors[src.t]
            h, w = op["hw"]
            xp, xbs, ldx = self._slice_ptr(self.tensors, src)
            yp = self.raw.data_ptr() + (op["level_off"] * self.rw + op["ch_off"]) * 4
            bias = self._padded_bias(op, name)
            self._conv_launch(xp, xbs, ldx, h, w, src.c, self.packed[name + ":fwd"], yp, self.A * self.rw, self.rw, h, w,
                              op["cout"], 1, 1, 0, bias=bias, out_f32=1)
        elif kind == "convt":
            name, src, dst = op["name"], op["src"], op["dst"]
            tin, tout = self.tensors[src.t], self.tensors[dst.t]
            h, w = tin.shape[1:3]
            xp, xbs, ldx = self._slice_ptr(self.tensors, src)
            yp, ybs, ldy = self._slice_ptr(self.tensors, dst)
            bias = self._padded_bias(op, name)
            self._conv_launch(xp, xbs, ldx, h, w, src.c, self.packed[name + ":fwd"], yp, ybs, ldy, h, w, 4 * dst.c, 1, 1, 0,
                              bias=bias, convt_co=dst.c)
        elif kind == "pool":
            src, dst = op["src"], op["dst"]
            t = self.tensors[src.t]
            _, H, W, Ct = t.shape
            xp, xbs, ldx = self._slice_ptr(self.tensors, src)
            yp, ybs, ldy = self._slice_ptr(self.tensors, dst)
            check(lib.m355_sppf_pool_launch(xp, xbs, ldx, yp, ybs, ldy, B, H, W, src.c, st))
        elif kind == "up":
            src, dst = op["src"], op["dst"]
            t = self.tensors[src.t]
            _, H, W, _ = t.shape
            xp, xbs, ldx = self._slice_ptr(self.tensors, src)
            yp, ybs, ldy = self._slice_ptr(self.tensors, dst)
            check(lib.m355_upsample2x_launch(xp, xbs, ldx, yp, ybs, ldy, B, H, W, src.c, st))
        elif kind == "addsilu":
            # RepConvN's tail: SiLU(a + b) of the two activation-free branches, one pass (csrc/train_kernels.hip: addsilu_fwd_kernel);
            # v = fp16(a + b) is kept for the backward pass
            a, b, dst = op["a"], op["b"], op["dst"]
            ta, tb = self.tensors[a.t], self.tensors[b.t]
            v = op.get("_v")
            if v is None:
                v = op["_v"] = torch.empty(ta.shape, dtype=torch.float16, device=self.dev)
            assert a.off == 0 and b.off == 0 and ta.shape[3] == a.c and tb.shape[3] == b.c
            yp, _, ldy = self._slice_ptr(self.tensors, dst)
            check(lib.m355_addsilu_fwd_launch(ta.data_ptr(), tb.data_ptr(), v.data_ptr(), yp, ta.shape[0] * ta.shape[1] * ta.shape[2], ldy, a.c, st))
        elif kind == "adown":
            # ADown's pooling front (2x2 average at stride 1, channel split, 3x3 / s2 max-pool of the second half) as one launch
            # (adown_fwd_kernel; the max-pool's argmax goes to a byte map for the backward); its two convolutions follow as conv ops
            src, p1, p2 = op["src"], op["p1"], op["p2"]
            _, H, W, _ = self.tensors[src.t].shape
            c = src.c // 2
            arg = op.get("_arg")
            if arg is None:
                arg = op["_arg"] = torch.empty((B, H // 2, W // 2, c), dtype=torch.uint8, device=self.dev)
            assert (H - 2) // 2 + 1 == H // 2 and (W - 2) // 2 + 1 == W // 2
            xp, xbs, ldx = self._slice_ptr(self.tensors, src)
            p1p, p1bs, ld1 = self._slice_ptr(self.tensors, p1)
            p2p, p2bs, ld2 = self._slice_ptr(self.tensors, p2)
            check(lib.m355_adown_fwd_launch(xp, xbs, ldx, p1p, p1bs, ld1, p2p, p2bs, ld2, arg.data_ptr(), B, H, W, c, st))

    def _tview(self, sl: Slice) -> torch.Tensor:
        return self.tensors[sl.t][..., sl.off:sl.off + sl.c]

    # ------------------------------------------------------------------ forward
    def forward(self, images_u8_nhwc: torch.Tensor, update_running_stats: bool = True):
        """images uint8 (B,H,W,3) on the device.  Returns raw (B,A,64+nc+32) fp32 and protos (B,H/4,W/4,32) fp16
        (views of engine buffers, valid until the next forward)."""
        B = self.B
        assert tuple(images_u8_nhwc.shape) == (B, *self.imgsz, 3) and images_u8_nhwc.dtype == torch.uint8
        x8 = self.tensors[self.x8]
        st = self._stream()
        # (u8 / 255) -> fp16 into the 8-channel input rows in one pass (was float(), div, half(), strided copy: 0.55 ms at b64)
        src = images_u8_nhwc.contiguous()
        check(lib.m355_u8_to_f16x8_launch(src.data_ptr(), x8.data_ptr(), x8.numel() // 8, st))
        side = self._head_stream
        if side is not None and self._head_ops is None and not any(o.get("name") == "model.15.cv2" for o in self.ops):
            side = self._head_stream = None                                 # (a graph without that fork point: one stream)
        if side is None:
            for op in self.ops:
                self._fwd_op(op, update_running_stats)
        else:
            # The head of the 1/8 level and the prototype branch (the large-map half of the head) need only model.15's output:
            # they go to a side stream while this stream walks the rest of the neck (40x40 / 20x20 maps that leave most CUs idle)
            # and the two smaller head levels.  The backward pass keeps the list order (its accumulations are ordered).
            if self._head_ops is None:
                self._head_ops = [o for o in self.ops if o.get("name", "").startswith(("model.22.cv2.0.", "model.22.cv3.0.",
                                                                                       "model.22.cv4.0.", "model.22.proto."))]
                for o in self._head_ops:
                    o["_side"] = True
                self._head_fork, self._head_join = torch.cuda.Event(), torch.cuda.Event()
            for op in self.ops:
                if op.get("_side"):
                    continue
                self._fwd_op(op, update_running_stats)
                if op.get("name") == "model.15.cv2":
                    self._head_fork.record(torch.cuda.current_stream())
                    side.wait_event(self._head_fork)
                    with torch.cuda.stream(side):
                        for hop in self._head_ops:
                            self._fwd_op(hop, update_running_stats)
                        self._head_join.record(side)
            torch.cuda.current_stream().wait_event(self._head_join)
        return self.raw, self.tensors[self.protos_t]

    # ------------------------------------------------------------------ backward
    def _gview(self, sl: Slice) -> torch.Tensor:
        return self.gtensors[sl.t][..., sl.off:sl.off + sl.c]

    def _uncovered(self, written, sl: Slice):
        """Channel intervals of a gradient slice nobody has written yet in this backward."""
        lo, hi = sl.off, sl.off + sl.c
        out, cur = [], lo
        for a, b in sorted(iv for iv in written.get(sl.t, []) if iv[0] < hi and iv[1] > lo):
            if a > cur:
                out.append((cur, a))
            cur = max(cur, b)
        if cur < hi:
            out.append((cur, hi))
        return out

    def _claim(self, written, sl: Slice) -> bool:
        """True when nothing of this gradient slice has been written yet: the caller STORES its contribution instead of
        accumulating it (no zero fill of the gradient buffers -- 100 fill kernels, 1.1 ms of a 46 ms step -- and the first
        dgrad into a slice does not read zeros back as its residual).  A partly written slice has its unwritten channels
        zeroed and accumulates."""
        gaps = self._uncovered(written, sl)
        first = gaps == [(sl.off, sl.off + sl.c)]
        if not first:
            for a, b in gaps:
                self.gtensors[sl.t][..., a:b].zero_()
        written.setdefault(sl.t, []).append((sl.off, sl.off + sl.c))
        return first

    def _ensure(self, written, sl: Slice) -> None:
        """A gradient slice about to be READ: channels no consumer wrote (none in the graphs built here) are zero."""
        for a, b in self._uncovered(written, sl):
            self.gtensors[sl.t][..., a:b].zero_()
        written.setdefault(sl.t, []).append((sl.off, sl.off + sl.c))

    def backward(self, d_raw: torch.Tensor, d_protos: torch.Tensor, on_ready=None) -> None:
        """d_raw (B,A,64+nc+32) fp32, d_protos (B,H/4,W/4,32): gradients of the loss w.r.t. forward()'s outputs.
        Fills ``self.grads`` (fp32, parameter layout).  ``on_ready(name)`` is called once the kernels producing that
        parameter's gradient have been enqueued (GradBucketReducer.mark_ready: the all-reduce is stream-ordered)."""
        B = self.B
        st = self._stream()
        d_raw = d_raw.float().contiguous()                                  # no-ops for the loss's own gradient buffer
        for i, t in enumerate(self.tensors):
            if self.gtensors[i] is None and i not in self.galias:
                self.gtensors[i] = torch.zeros_like(t)
        for i, j in self.galias.items():                                    # RepConvN branches: one gradient buffer for both
            self.gtensors[i] = self.gtensors[j]
        written: Dict[int, list] = {}                                     # tensor -> channel intervals written in this backward
        level_bias: Dict[int, torch.Tensor] = {}                          # level offset -> per-channel sums of d_raw over the level
        self.gtensors[self.protos_t].copy_(d_protos.to(torch.float16))
        written[self.protos_t] = [(0, self.tensors[self.protos_t].shape[-1])]
        ready: List[str] = []

        def run(op):
            nonlocal ready
            st = self._stream()                                             # (the stream this op is enqueued on)
            if on_ready is not None and ready:                              # gradients finished by the previous op
                self._join_side()                                           # (its weight gradient ran on the side stream)
                for k in ready:
                    on_ready(k)
            ready = []
            kind = op["kind"]
            if kind == "conv":
                name, src, dst = op["name"], op["src"], op["dst"]
                s = self.specs[name]
                sv = self.saved[name]
                tin, tout = self.tensors[src.t], self.tensors[dst.t]
                hi, wi = tin.shape[1:3]
                ho, wo = tout.shape[1:3]
                cout, cin = dst.c, src.c
                self._ensure(written, dst)
                if op["res"] is not None:                                   # y = act(bn(z)) + res
                    if self._claim(written, op["res"]):
                        self._gview(op["res"]).copy_(self._gview(dst))
                    else:
                        self._gview(op["res"]).add_(self._gview(dst))
                dyp, _, lddy = self._slice_ptr(self.gtensors, dst)
                if "dz" not in sv:
                    sv["dz"] = torch.empty_like(sv["z"])
                gb = self.grads[f"{name}.bn.bias"]                        # [dbeta | dgamma] land in the flat buffer
                assert self.grads[f"{name}.bn.weight"].data_ptr() == gb.data_ptr() + 4 * cout
                check(lib.m355_bn_train_bwd_launch(sv["z"].data_ptr(), dyp, B * ho * wo, cout, lddy, cout,
                                                   sv["mean"].data_ptr(), sv["invstd"].data_ptr(),
                                                   self.params[f"{name}.bn.weight"].data_ptr(),
                                                   self.params[f"{name}.bn.bias"].data_ptr(), op.get("act", 1), sv["dz"].data_ptr(), cout,
                                                   gb.data_ptr(), sv["ws"].data_ptr(), st))
                ready += [f"{name}.bn.bias", f"{name}.bn.weight", f"{name}.conv.weight"]
                xp, xbs, ldx = self._slice_ptr(self.tensors, src)
                gw = self.grads[f"{name}.conv.weight"]
                if s.cin == 3:
                    if "dw8" not in sv:
                        sv["dw8"] = torch.empty((cout, 3, 3, 8), device=self.dev)
                    with self._beside(op):      # (every weight gradient on the one side stream: they share the split-K workspace)
                        self._wgrad_launch(sv["dz"].data_ptr(), cout, ho * wo * cout, xp, xbs, ldx, hi, wi, 8, ho, wo, cout, s.k,
                                           s.stride, s.k // 2, sv["dw8"])
                        gw.copy_(sv["dw8"][..., :3])
                else:
                    with self._beside(op):
                        self._wgrad_launch(sv["dz"].data_ptr(), cout, ho * wo * cout, xp, xbs, ldx, hi, wi, cin, ho, wo, cout, s.k,
                                           s.stride, s.k // 2, gw)
                    gp, gbs, ldg = self._slice_ptr(self.gtensors, src)
                    acc = 0 if self._claim(written, src) else gp
                    if (s.stride == 2 and s.k == 3 and self._dgrad_phases and hi == 2 * ho and wi == 2 * wo
                            and (cin % 64 == 0 or (128 % cin == 0 and not acc))):
                        self._conv_launch(sv["dz"].data_ptr(), ho * wo * cout, cout, ho, wo, cout, self.packed[name + ":dgrad4"],
                                          gp, gbs, ldg, ho, wo, 4 * cin, 2, 1, 0, res_ptr=acc, r_bs=gbs, ldr=ldg, convt_co=cin, tmode=2)
                    else:
                        self._conv_launch(sv["dz"].data_ptr(), ho * wo * cout, cout, ho, wo, cout, self.packed[name + ":dgrad"],
                                          gp, gbs, ldg, hi, wi, cin, s.k, 1, s.k // 2, res_ptr=acc, r_bs=gbs, ldr=ldg,
                                          tmode=1 if s.stride == 2 else 0)
            elif kind == "plain":
                name, src = op["name"], op["src"]
                s = self.specs[name]
                h, w = op["hw"]
                cout, cp = op["cout"], _ceil(op["cout"], 8)
                lo = op["level_off"]
                # fp16 copy of this conv's slice of d_raw in a persistent buffer (its padding channels stay zero); the bias
                # gradients of a level's three output convs are ONE fp32 reduction over the level's rows of d_raw
                dz = op.get("_dz")
                if dz is None:
                    dz = op["_dz"] = torch.zeros((B, h * w, cp), dtype=torch.float16, device=self.dev)
                dz[..., :cout].copy_(d_raw[:, lo:lo + h * w, op["ch_off"]:op["ch_off"] + cout])
                if lo not in level_bias and d_raw.shape[2] > 256:           # (more than 160 classes: wider than the column-sum kernel)
                    level_bias[lo] = d_raw[:, lo:lo + h * w, :].sum((0, 1))
                if lo not in level_bias:
                    level_bias[lo] = self._colsum(d_raw.data_ptr() + lo * d_raw.shape[2] * 4, False, B, d_raw.shape[1] * d_raw.shape[2],
                                                  h * w, d_raw.shape[2], d_raw.shape[2], torch.empty(d_raw.shape[2], device=self.dev))
                self.grads[f"{name}.bias"].copy_(level_bias[lo][op["ch_off"]:op["ch_off"] + cout])
                xp, xbs, ldx = self._slice_ptr(self.tensors, src)
                gw = self.grads[f"{name}.weight"]
                dw = op.get("_dw")
                if cp != cout and dw is None:
                    dw = op["_dw"] = torch.empty((cp, 1, 1, src.c), device=self.dev)
                with self._beside(op):
                    if cp == cout:                                          # KRSC rows = the gradient's own layout: no staging copy
                        self._wgrad_launch(dz.data_ptr(), cp, h * w * cp, xp, xbs, ldx, h, w, src.c, h, w, cp, 1, 1, 0, gw)
                    else:
                        self._wgrad_launch(dz.data_ptr(), cp, h * w * cp, xp, xbs, ldx, h, w, src.c, h, w, cp, 1, 1, 0, dw)
                        gw.copy_(dw[:cout])
                ready += [f"{name}.bias", f"{name}.weight"]
                gp, gbs, ldg = self._slice_ptr(self.gtensors, src)
                acc = 0 if self._claim(written, src) else gp
                self._conv_launch(dz.data_ptr(), h * w * cp, cp, h, w, cp, self.packed[name + ":dgrad"], gp, gbs, ldg, h, w,
                                  src.c, 1, 1, 0, res_ptr=acc, r_bs=gbs, ldr=ldg)
            elif kind == "convt":
                name, src, dst = op["name"], op["src"], op["dst"]
                tin = self.tensors[src.t]
                h, w = tin.shape[1:3]
                cin, cout = src.c, dst.c
                self._ensure(written, dst)
                gy = self.gtensors[dst.t]                                   # (B, 2h, 2w, cout) contiguous, own tensor
                self._colsum(gy.data_ptr(), True, 1, 0, B * 4 * h * w, cout, cout, self.grads[f"{name}.bias"])
                xp, xbs, ldx = self._slice_ptr(self.tensors, src)
                # wgrad of the equivalent 2x2 / stride-2 conv (dY -> X): "dz" = X, "x" = dY -> [cin][(dy,dx),co]
                dw = op.get("_dw")
                if dw is None:
                    dw = op["_dw"] = torch.empty((cin, 2, 2, cout), device=self.dev)
                with self._beside(op):
                    self._wgrad_launch(xp, ldx, xbs, gy.data_ptr(), 4 * h * w * cout, cout, 2 * h, 2 * w, cout, h, w, cin, 2, 2, 0, dw)
                    self.grads[f"{name}.weight"].copy_(dw.permute(0, 3, 1, 2))
                ready += [f"{name}.bias", f"{name}.weight"]
                gp, gbs, ldg = self._slice_ptr(self.gtensors, src)
                acc = 0 if self._claim(written, src) else gp
                self._conv_launch(gy.data_ptr(), 4 * h * w * cout, cout, 2 * h, 2 * w, cout, self.packed[name + ":dgrad"], gp,
                                  gbs, ldg, h, w, cin, 2, 2, 0, res_ptr=acc, r_bs=gbs, ldr=ldg)
            elif kind == "pool":                                           # SPPF: y1 = mp(a), y2 = mp(y1), y3 = mp(y2)
                src, dst = op["src"], op["dst"]
                c = src.c
                self._ensure(written, Slice(dst.t, dst.off, 3 * c))
                hp, wp = self.tensors[src.t].shape[1:3]
                first = self._claim(written, src)
                if hp * wp * 96 <= 160 * 1024:                              # the map of 8 channels fits one block's LDS
                    ap, abs_, lda = self._slice_ptr(self.tensors, src)
                    yp, ybs, ldy = self._slice_ptr(self.tensors, dst)
                    gyp, gybs, ldgy = self._slice_ptr(self.gtensors, dst)
                    gap, gabs, ldga = self._slice_ptr(self.gtensors, src)
                    check(lib.m355_sppf_pool_bwd_launch(ap, abs_, lda, yp, ybs, ldy, gyp, gybs, ldgy, gap, gabs, ldga, B, hp, wp, c,
                                                        0 if first else 1, st))
                else:   # larger maps (network input > 1300 px): torch's pooling on contiguous NCHW fp32 copies
                    a = self.tensors[src.t][..., src.off:src.off + c].permute(0, 3, 1, 2).float().contiguous().requires_grad_(True)
                    y1 = F.max_pool2d(a, 5, 1, 2)
                    y2 = F.max_pool2d(y1, 5, 1, 2)
                    y3 = F.max_pool2d(y2, 5, 1, 2)
                    g = self.gtensors[dst.t][..., dst.off:dst.off + 3 * c].permute(0, 3, 1, 2).float().contiguous()
                    (ga,) = torch.autograd.grad((y1, y2, y3), a, (g[:, :c], g[:, c:2 * c], g[:, 2 * c:]))
                    if first:
                        self._gview(src).copy_(ga.permute(0, 2, 3, 1))
                    else:
                        self._gview(src).add_(ga.permute(0, 2, 3, 1).half())
            elif kind == "addsilu":                                        # y = SiLU(v), v = a + b: dv = dy * SiLU'(v) for BOTH branches
                a_, b_, dst = op["a"], op["b"], op["dst"]
                self._ensure(written, dst)
                gyp, _, ldgy = self._slice_ptr(self.gtensors, dst)
                g = self.gtensors[a_.t]
                check(lib.m355_addsilu_bwd_launch(op["_v"].data_ptr(), gyp, ldgy, g.data_ptr(), g.shape[0] * g.shape[1] * g.shape[2], a_.c, st))
                written[a_.t] = [(0, a_.c)]
                written[b_.t] = [(0, b_.c)]                                 # (the same buffer: galias)
            elif kind == "adown":                                          # gather backward of the pooling front (adown_bwd_kernel)
                src, p1, p2 = op["src"], op["p1"], op["p2"]
                self._ensure(written, p1)
                self._ensure(written, p2)
                _, H, W, _ = self.tensors[src.t].shape
                g1p, g1bs, ld1 = self._slice_ptr(self.gtensors, p1)
                g2p, g2bs, ld2 = self._slice_ptr(self.gtensors, p2)
                gxp, gxbs, ldg = self._slice_ptr(self.gtensors, src)
                check(lib.m355_adown_bwd_launch(g1p, g1bs, ld1, g2p, g2bs, ld2, op["_arg"].data_ptr(), gxp, gxbs, ldg, B, H, W, src.c // 2,
                                                0 if self._claim(written, src) else 1, st))
            elif kind == "up":
                src, dst = op["src"], op["dst"]
                self._ensure(written, dst)
                hs, ws_ = self.tensors[src.t].shape[1:3]
                gyp, gybs, ldgy = self._slice_ptr(self.gtensors, dst)
                gp, gbs, ldg = self._slice_ptr(self.gtensors, src)
                check(lib.m355_upsample2x_bwd_launch(gyp, gybs, ldgy, gp, gbs, ldg, B, hs, ws_, src.c,
                                                     0 if self._claim(written, src) else 1, st))

        ops_rev = list(reversed(self.ops))
        side = self._head_stream if (on_ready is None and self._head_ops) else None
        if side is None:
            for op in ops_rev:
                run(op)
        else:
            # The backward of the 1/8-level head + prototype branch (large maps) on the head stream beside the backward of the two
            # smaller head levels on this one: they write disjoint gradient tensors (model.15's output vs model.18's / model.21's)
            # and each stream keeps the list order among its own ops, so every accumulation happens in the order of the
            # one-stream pass.  Joined before the neck's backward starts.  (One stream when a gradient reducer is attached.)
            head_rest = [o for o in ops_rev if o.get("name", "").startswith("model.22.") and not o.get("_side")]
            self._head_fork.record(torch.cuda.current_stream())
            side.wait_event(self._head_fork)
            with torch.cuda.stream(side):
                for op in ops_rev:
                    if op.get("_side"):
                        run(op)
                self._head_join.record(side)
            for op in head_rest:
                run(op)
            torch.cuda.current_stream().wait_event(self._head_join)
            for op in ops_rev:
                if not op.get("name", "").startswith("model.22."):
                    run(op)
        self._join_side()
        if on_ready is not None:
            for k in ready:
                on_ready(k)
