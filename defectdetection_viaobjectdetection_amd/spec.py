"""Host-side description of the YOLOv8-seg parameter set (names, shapes, initialisation, BN folding).

Mirrors what ``ultralytics.YOLO("yolov8{n,s,m,l,x}-seg.yaml")`` builds
(/root/reference/BscanBased/yolo_seg_train.py:7): the same state-dict
key names, so weights saved by this package load by name and an upstream state dict maps 1:1.
The canonical conv order here equals the order ``libmi355yolo`` reports through
``m355_get_conv_info`` (checked by tests/test_engine_gpu.py).

PyTorch is used for tensors and RNG only -- no network arithmetic happens in this file.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List

import torch

SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768),
          "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512)}
V9C = "9c"     # the `scale` tag of the yolov9c-seg graph (SURVEY next row N4); the C-ABI descriptor carries it as 'c'
REG_MAX = 16
NM = 32
BN_EPS = 1e-3


@dataclass(frozen=True)
class ConvSpec:
    name: str          # state-dict prefix
    cin: int
    cout: int
    k: int
    stride: int
    has_bn: bool       # Conv2d(bias=False)+BN+SiLU  vs plain Conv2d/ConvTranspose2d with bias
    transposed: bool = False
    rep: bool = False  # RepConvN: act(Conv3x3+BN [name.conv1] + Conv1x1+BN [name.conv2]); the engine runs the merged 3x3

    @property
    def weight_shape(self):
        return (self.cin, self.cout, 2, 2) if self.transposed else (self.cout, self.cin, self.k, self.k)


def _make_divisible(x: float, d: int) -> int:
    return int(math.ceil(x / d) * d)


def conv_specs_v9c(nc: int = 1) -> List[ConvSpec]:
    """Canonical list of every convolution of yolov9c-seg (row N4; /root/reference/BscanBased/yolo_seg_train.py:7), in the
    order ``libmi355yolo`` reports them.  Block structure: oracle/yolov9c_seg_oracle.py (exact published parameter counts)."""
    out: List[ConvSpec] = []

    def conv(name, cin, cout, k, s):
        out.append(ConvSpec(name, cin, cout, k, s, True))

    def repcsp(name, c1, c2):
        c_ = c2 // 2
        conv(f"{name}.cv1", c1, c_, 1, 1)
        out.append(ConvSpec(f"{name}.m.0.cv1", c_, c_, 3, 1, True, False, True))
        conv(f"{name}.m.0.cv2", c_, c_, 3, 1)
        conv(f"{name}.cv2", c1, c_, 1, 1)
        conv(f"{name}.cv3", 2 * c_, c2, 1, 1)

    def elan(name, c1, c2, c3, c4):
        conv(f"{name}.cv1", c1, c3, 1, 1)
        repcsp(f"{name}.cv2.0", c3 // 2, c4)
        conv(f"{name}.cv2.1", c4, c4, 3, 1)
        repcsp(f"{name}.cv3.0", c4, c4)
        conv(f"{name}.cv3.1", c4, c4, 3, 1)
        conv(f"{name}.cv4", c3 + 2 * c4, c2, 1, 1)

    def adown(name, c1, c2):
        conv(f"{name}.cv1", c1 // 2, c2 // 2, 3, 2)
        conv(f"{name}.cv2", c1 // 2, c2 // 2, 1, 1)

    conv("model.0", 3, 64, 3, 2)
    conv("model.1", 64, 128, 3, 2)
    elan("model.2", 128, 256, 128, 64)
    adown("model.3", 256, 256)
    elan("model.4", 256, 512, 256, 128)
    adown("model.5", 512, 512)
    elan("model.6", 512, 512, 512, 256)
    adown("model.7", 512, 512)
    elan("model.8", 512, 512, 512, 256)
    conv("model.9.cv1", 512, 256, 1, 1)
    conv("model.9.cv5", 1024, 512, 1, 1)
    elan("model.12", 1024, 512, 512, 256)
    elan("model.15", 1024, 256, 256, 128)
    adown("model.16", 256, 256)
    elan("model.18", 768, 512, 512, 256)
    adown("model.19", 512, 512)
    elan("model.21", 1024, 512, 512, 256)
    _segment_specs(out, nc, (256, 512, 512), 256)
    return out


def _segment_specs(out: List[ConvSpec], nc: int, fch, npr: int) -> None:
    """model.22 = Segment: the box / class / coefficient branches per level and Proto (A9/A10), upstream state-dict order."""
    hc2 = max(16, fch[0] // 4, REG_MAX * 4)
    hc3 = max(fch[0], min(nc, 100))
    hc4 = max(fch[0] // 4, NM)

    def conv(name, cin, cout, k, s):
        out.append(ConvSpec(name, cin, cout, k, s, True))
    for l in range(3):
        conv(f"model.22.cv2.{l}.0", fch[l], hc2, 3, 1)
        conv(f"model.22.cv2.{l}.1", hc2, hc2, 3, 1)
        out.append(ConvSpec(f"model.22.cv2.{l}.2", hc2, 4 * REG_MAX, 1, 1, False))
    for l in range(3):
        conv(f"model.22.cv3.{l}.0", fch[l], hc3, 3, 1)
        conv(f"model.22.cv3.{l}.1", hc3, hc3, 3, 1)
        out.append(ConvSpec(f"model.22.cv3.{l}.2", hc3, nc, 1, 1, False))
    conv("model.22.proto.cv1", fch[0], npr, 3, 1)
    out.append(ConvSpec("model.22.proto.upsample", npr, npr, 2, 2, False, True))
    conv("model.22.proto.cv2", npr, npr, 3, 1)
    conv("model.22.proto.cv3", npr, NM, 1, 1)
    for l in range(3):
        conv(f"model.22.cv4.{l}.0", fch[l], hc4, 3, 1)
        conv(f"model.22.cv4.{l}.1", hc4, hc4, 3, 1)
        out.append(ConvSpec(f"model.22.cv4.{l}.2", hc4, NM, 1, 1, False))


def conv_specs(scale: str = "s", nc: int = 1) -> List[ConvSpec]:
    """Canonical list of every convolution of yolov8{scale}-seg (A5/A9/A10); scale "9c": yolov9c-seg."""
    if scale == V9C:
        return conv_specs_v9c(nc)
    depth, width, maxc = SCALES[scale]
    ch = lambda c: _make_divisible(min(c, maxc) * width, 8)  # noqa: E731
    rep = lambda n: max(round(n * depth), 1) if n > 1 else n  # noqa: E731
    c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    out: List[ConvSpec] = []

    def conv(name, cin, cout, k, s):
        out.append(ConvSpec(name, cin, cout, k, s, True))

    def c2f(name, cin, cout, n):
        c = cout // 2
        conv(f"{name}.cv1", cin, 2 * c, 1, 1)
        for j in range(n):
            conv(f"{name}.m.{j}.cv1", c, c, 3, 1)
            conv(f"{name}.m.{j}.cv2", c, c, 3, 1)
        conv(f"{name}.cv2", (2 + n) * c, cout, 1, 1)

    conv("model.0", 3, c64, 3, 2)
    conv("model.1", c64, c128, 3, 2)
    c2f("model.2", c128, c128, rep(3))
    conv("model.3", c128, c256, 3, 2)
    c2f("model.4", c256, c256, rep(6))
    conv("model.5", c256, c512, 3, 2)
    c2f("model.6", c512, c512, rep(6))
    conv("model.7", c512, c1024, 3, 2)
    c2f("model.8", c1024, c1024, rep(3))
    conv("model.9.cv1", c1024, c1024 // 2, 1, 1)
    conv("model.9.cv2", c1024 * 2, c1024, 1, 1)
    c2f("model.12", c1024 + c512, c512, rep(3))
    c2f("model.15", c512 + c256, c256, rep(3))
    conv("model.16", c256, c256, 3, 2)
    c2f("model.18", c256 + c512, c512, rep(3))
    conv("model.19", c512, c512, 3, 2)
    c2f("model.21", c512 + c1024, c1024, rep(3))
    fch = (c256, c512, c1024)
    hc2 = max(16, fch[0] // 4, REG_MAX * 4)
    hc3 = max(fch[0], min(nc, 100))
    hc4 = max(fch[0] // 4, NM)
    npr = ch(256)
    for l in range(3):
        conv(f"model.22.cv2.{l}.0", fch[l], hc2, 3, 1)
        conv(f"model.22.cv2.{l}.1", hc2, hc2, 3, 1)
        out.append(ConvSpec(f"model.22.cv2.{l}.2", hc2, 4 * REG_MAX, 1, 1, False))
    for l in range(3):
        conv(f"model.22.cv3.{l}.0", fch[l], hc3, 3, 1)
        conv(f"model.22.cv3.{l}.1", hc3, hc3, 3, 1)
        out.append(ConvSpec(f"model.22.cv3.{l}.2", hc3, nc, 1, 1, False))
    conv("model.22.proto.cv1", fch[0], npr, 3, 1)
    out.append(ConvSpec("model.22.proto.upsample", npr, npr, 2, 2, False, True))
    conv("model.22.proto.cv2", npr, npr, 3, 1)
    conv("model.22.proto.cv3", npr, NM, 1, 1)
    for l in range(3):
        conv(f"model.22.cv4.{l}.0", fch[l], hc4, 3, 1)
        conv(f"model.22.cv4.{l}.1", hc4, hc4, 3, 1)
        out.append(ConvSpec(f"model.22.cv4.{l}.2", hc4, NM, 1, 1, False))
    return out


def conv_branches(s: ConvSpec):
    """(state-dict prefix, kernel size) of the Conv2d+BN pairs behind one engine conv: itself, or RepConvN's two branches."""
    return [(f"{s.name}.conv1", 3), (f"{s.name}.conv2", 1)] if s.rep else [(s.name, s.k)]


def state_dict_keys(scale: str, nc: int) -> List[str]:
    keys = []
    for s in conv_specs(scale, nc):
        if s.has_bn:
            for pre, _ in conv_branches(s):
                keys += [f"{pre}.conv.weight"] + [f"{pre}.bn.{p}" for p in
                                                  ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
        else:
            keys += [f"{s.name}.weight", f"{s.name}.bias"]
    keys.append("model.22.dfl.conv.weight")
    return keys


def count_parameters(sd: Dict[str, torch.Tensor]) -> int:
    """Learnable + frozen parameters as upstream counts them (BN buffers excluded)."""
    return sum(v.numel() for k, v in sd.items()
               if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))


def _bias_init(sd: Dict[str, torch.Tensor], nc: int, imgsz: int = 640) -> None:
    """A.1 head bias init: box branch 1.0, class branch log(5/nc/(imgsz/stride)^2)."""
    for l, s in enumerate((8, 16, 32)):
        sd[f"model.22.cv2.{l}.2.bias"].fill_(1.0)
        sd[f"model.22.cv3.{l}.2.bias"][:nc] = math.log(5 / nc / (imgsz / s) ** 2)


def init_state_dict(scale: str = "s", nc: int = 1, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Fresh weights as ``YOLO("*.yaml")`` would create them: PyTorch-default conv init
    (kaiming_uniform(a=sqrt 5) -> U(+-1/sqrt(fan_in))), BN gamma=1 beta=0 stats (0,1), head biases A.1."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for s in conv_specs(scale, nc):
        shape = s.weight_shape
        fan_in = shape[1] * shape[2] * shape[3]
        bound = 1.0 / math.sqrt(fan_in)
        w = (torch.rand(shape, generator=g) * 2 - 1) * bound
        if s.has_bn:
            for pre, k in conv_branches(s):
                if s.rep:
                    fi = s.cin * k * k
                    w = (torch.rand((s.cout, s.cin, k, k), generator=g) * 2 - 1) / math.sqrt(fi)
                sd[f"{pre}.conv.weight"] = w
                sd[f"{pre}.bn.weight"] = torch.ones(s.cout)
                sd[f"{pre}.bn.bias"] = torch.zeros(s.cout)
                sd[f"{pre}.bn.running_mean"] = torch.zeros(s.cout)
                sd[f"{pre}.bn.running_var"] = torch.ones(s.cout)
                sd[f"{pre}.bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        else:
            sd[f"{s.name}.weight"] = w
            sd[f"{s.name}.bias"] = (torch.rand(s.cout, generator=g) * 2 - 1) * bound
    sd["model.22.dfl.conv.weight"] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
    _bias_init(sd, nc)
    return sd


def _load_gains(scale: str) -> Dict[str, float]:
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"synth_gains_{scale}.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)["gains"]


def synthetic_state_dict(scale: str = "s", nc: int = 1, seed: int = 0, cls_bias: float = -3.5,
                         gains: Dict[str, float] = None) -> Dict[str, torch.Tensor]:
    """Seeded *scale-calibrated* weights for parity tests and the synthetic benchmark (SURVEY 8d
    config 2).  conv ~ U(+-g*sqrt(3/fan_in)) where the per-layer gain g comes from
    ``data/synth_gains_{scale}.json`` -- measured once so that every conv output has unit standard
    deviation on synthetic B-scans (script: tests/golden/make_synth_gains.py; without the file g = 1.67,
    the analytic SiLU compensation).  BN gets a non-trivial affine + running statistics so that folding is
    exercised; the box-branch bias is 1.0 and the class bias lets on the order of 1 % of the anchors
    pass conf = 0.25, so NMS and mask assembly do real work (the default head-bias init would leave no
    detection at all)."""
    if gains is None:
        gains = _load_gains(scale)
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for s in conv_specs(scale, nc):
        shape = s.weight_shape
        fan_in = (shape[0] if s.transposed else shape[1] * shape[2] * shape[3])
        gain = gains.get(s.name, 1.67 if s.has_bn else 1.0)
        bound = gain * math.sqrt(3.0 / fan_in)
        w = (torch.rand(shape, generator=g) * 2 - 1) * bound
        if s.has_bn:
            for pre, k in conv_branches(s):
                if s.rep:     # the two branches add up: each gets 1/sqrt(2) of the calibrated gain (one gain per RepConvN)
                    w = (torch.rand((s.cout, s.cin, k, k), generator=g) * 2 - 1) * gain * math.sqrt(1.5 / (s.cin * k * k))
                sd[f"{pre}.conv.weight"] = w
                sd[f"{pre}.bn.weight"] = 0.8 + 0.4 * torch.rand(s.cout, generator=g)
                sd[f"{pre}.bn.bias"] = 0.2 * torch.rand(s.cout, generator=g) - 0.1
                sd[f"{pre}.bn.running_mean"] = 0.2 * torch.rand(s.cout, generator=g) - 0.1
                sd[f"{pre}.bn.running_var"] = 0.8 + 0.4 * torch.rand(s.cout, generator=g)
                sd[f"{pre}.bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        else:
            sd[f"{s.name}.weight"] = w
            sd[f"{s.name}.bias"] = 0.2 * torch.rand(s.cout, generator=g) - 0.1
    sd["model.22.dfl.conv.weight"] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
    for l in range(3):
        sd[f"model.22.cv2.{l}.2.bias"].fill_(1.0)
        sd[f"model.22.cv3.{l}.2.bias"].fill_(cls_bias)
    return sd


def fold_bn(sd: Dict[str, torch.Tensor], spec: ConvSpec):
    """A4: W' = W * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var + eps).
    Returns (weight fp32 contiguous, bias fp32 contiguous) ready for ``m355_set_conv_weights``."""
    if spec.rep:   # RepConvN: both branches folded, the 1x1 kernel added at the centre tap of the 3x3 (exact: conv is linear)
        wsum = torch.zeros((spec.cout, spec.cin, 3, 3), dtype=torch.float64)
        bsum = torch.zeros(spec.cout, dtype=torch.float64)
        for pre, k in conv_branches(spec):
            scale = sd[f"{pre}.bn.weight"].double() / torch.sqrt(sd[f"{pre}.bn.running_var"].double() + BN_EPS)
            wk = sd[f"{pre}.conv.weight"].double() * scale.view(-1, 1, 1, 1)
            if k == 3:
                wsum += wk
            else:
                wsum[:, :, 1:2, 1:2] += wk
            bsum += sd[f"{pre}.bn.bias"].double() - sd[f"{pre}.bn.running_mean"].double() * scale
        return wsum.float().contiguous(), bsum.float().contiguous()
    if spec.has_bn:
        w = sd[f"{spec.name}.conv.weight"].double()
        gamma = sd[f"{spec.name}.bn.weight"].double()
        beta = sd[f"{spec.name}.bn.bias"].double()
        mean = sd[f"{spec.name}.bn.running_mean"].double()
        var = sd[f"{spec.name}.bn.running_var"].double()
        scale = gamma / torch.sqrt(var + BN_EPS)
        wf = (w * scale.view(-1, 1, 1, 1)).float().contiguous()
        bf = (beta - mean * scale).float().contiguous()
        return wf, bf
    return sd[f"{spec.name}.weight"].float().contiguous(), sd[f"{spec.name}.bias"].float().contiguous()
