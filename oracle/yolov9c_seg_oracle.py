"""CPU ORACLE for the YOLOv9c-seg graph  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` (and ``tools/`` diagnostics) may import this module.

The reference's training / prediction scripts literally name this architecture:
``YOLO("yolov9c-seg.yaml")`` /root/reference/BscanBased/yolo_seg_train.py:7, ``yolo9c-seg/.../best.pt``
/root/reference/BscanBased/yolo8_seg_predict.py:4.  Its arithmetic lives in the un-vendored ``ultralytics`` package
(``yolov9c-seg.yaml``: GELAN blocks of the YOLOv9 paper, arXiv 2402.13616); nothing of it exists under /root/reference.

PARITY STATUS: **parity unpinned at the reference level** (same situation as yolov8_seg_oracle.py).  This file restates
the published architecture in plain PyTorch-CPU fp32 and is pinned by
  (i)  EXACT parameter counts against the two published model summaries: 25 590 912 for yolov9c (detect head, nc=80)
       and 27 897 120 for yolov9c-seg (nc=80) -- every layer's shape is therefore right;
  (ii) output shapes (B, 4+nc+32, 8400) / (B, 32, 160, 160) at 640x640 and the module-level known answers in
       tests/test_v9c_oracle.py (ADown's pooling arithmetic, RepConvN = one 3x3 conv after branch fusion, SPPELAN's
       serial pooling).
The dataflow inside the blocks (what is concatenated with what) follows the paper's GELAN figure and the block
definitions upstream publishes; it cannot be checked against upstream outputs here.

Blocks (state-dict names in brackets):
  RepConvN        act(Conv3x3+BN [conv1] + Conv1x1+BN [conv2])                       (no identity branch)
  RepBottleneck   x + Conv3x3(RepConvN(x)) [cv1, cv2]
  RepCSP          cv3(cat(m(cv1(x)), cv2(x))), cv1 / cv2 / cv3 1x1, hidden = c2 / 2, m = n RepBottlenecks
  RepNCSPELAN4    y = chunk2(cv1(x)); y += [cv2(y[-1])]; y += [cv3(y[-1])]; cv4(cat(y)),
                  cv2 / cv3 = Sequential(RepCSP, Conv3x3)
  ADown           x = avg_pool2d(x, 2, 1, 0); x1, x2 = chunk2(x); cat(Conv3x3/s2(x1) [cv1], Conv1x1(max_pool2d(x2, 3, 2, 1)) [cv2])
  SPPELAN         y = [cv1(x)]; three serial MaxPool2d(5, 1, 2); cv5(cat(y))
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from yolov8_seg_oracle import NM, Conv, Segment


class ConvNoAct(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d, no activation (the branches of RepConvN)."""

    def __init__(self, c1: int, c2: int, k: int):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, 1, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)

    def forward(self, x):
        return self.bn(self.conv(x))


class RepConvN(nn.Module):
    def __init__(self, c1: int, c2: int):
        super().__init__()
        self.conv1 = ConvNoAct(c1, c2, 3)
        self.conv2 = ConvNoAct(c1, c2, 1)

    def forward(self, x):
        return F.silu(self.conv1(x) + self.conv2(x))


class RepBottleneck(nn.Module):
    def __init__(self, c1: int, c2: int):
        super().__init__()
        self.cv1 = RepConvN(c1, c2)
        self.cv2 = Conv(c2, c2, 3, 1)
        self.add = c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class RepCSP(nn.Module):
    def __init__(self, c1: int, c2: int, n: int = 1):
        super().__init__()
        c_ = int(c2 * 0.5)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1, 1)
        self.m = nn.Sequential(*(RepBottleneck(c_, c_) for _ in range(n)))

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class RepNCSPELAN4(nn.Module):
    def __init__(self, c1: int, c2: int, c3: int, c4: int, n: int = 1):
        super().__init__()
        self.c = c3 // 2
        self.cv1 = Conv(c1, c3, 1, 1)
        self.cv2 = nn.Sequential(RepCSP(c3 // 2, c4, n), Conv(c4, c4, 3, 1))
        self.cv3 = nn.Sequential(RepCSP(c4, c4, n), Conv(c4, c4, 3, 1))
        self.cv4 = Conv(c3 + 2 * c4, c2, 1, 1)

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        y.append(self.cv2(y[-1]))
        y.append(self.cv3(y[-1]))
        return self.cv4(torch.cat(y, 1))


class ADown(nn.Module):
    def __init__(self, c1: int, c2: int):
        super().__init__()
        self.c = c2 // 2
        self.cv1 = Conv(c1 // 2, self.c, 3, 2)
        self.cv2 = Conv(c1 // 2, self.c, 1, 1)

    def forward(self, x):
        x = F.avg_pool2d(x, 2, 1, 0, False, True)
        x1, x2 = x.chunk(2, 1)
        x1 = self.cv1(x1)
        x2 = self.cv2(F.max_pool2d(x2, 3, 2, 1))
        return torch.cat((x1, x2), 1)


class SPPELAN(nn.Module):
    def __init__(self, c1: int, c2: int, c3: int, k: int = 5):
        super().__init__()
        self.cv1 = Conv(c1, c3, 1, 1)
        self.cv5 = Conv(4 * c3, c2, 1, 1)
        self.k = k

    def forward(self, x):
        y = [self.cv1(x)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], self.k, 1, self.k // 2))
        return self.cv5(torch.cat(y, 1))


class SegmentationModelV9c(nn.Module):
    """The 23-entry yolov9c-seg graph; layer i is ``self.model[i]`` (upstream state-dict naming)."""

    def __init__(self, nc: int = 1):
        super().__init__()
        m: List[nn.Module] = [
            Conv(3, 64, 3, 2),                                # 0
            Conv(64, 128, 3, 2),                              # 1
            RepNCSPELAN4(128, 256, 128, 64, 1),               # 2
            ADown(256, 256),                                  # 3
            RepNCSPELAN4(256, 512, 256, 128, 1),              # 4
            ADown(512, 512),                                  # 5
            RepNCSPELAN4(512, 512, 512, 256, 1),              # 6
            ADown(512, 512),                                  # 7
            RepNCSPELAN4(512, 512, 512, 256, 1),              # 8
            SPPELAN(512, 512, 256),                           # 9
            nn.Upsample(scale_factor=2, mode="nearest"),      # 10
            nn.Identity(),                                    # 11 Concat[-1, 6]
            RepNCSPELAN4(1024, 512, 512, 256, 1),             # 12
            nn.Upsample(scale_factor=2, mode="nearest"),      # 13
            nn.Identity(),                                    # 14 Concat[-1, 4]
            RepNCSPELAN4(1024, 256, 256, 128, 1),             # 15
            ADown(256, 256),                                  # 16
            nn.Identity(),                                    # 17 Concat[-1, 12]
            RepNCSPELAN4(768, 512, 512, 256, 1),              # 18
            ADown(512, 512),                                  # 19
            nn.Identity(),                                    # 20 Concat[-1, 9]
            RepNCSPELAN4(1024, 512, 512, 256, 1),             # 21
            Segment(nc, NM, 256, (256, 512, 512)),            # 22
        ]
        self.model = nn.ModuleList(m)
        self.nc = nc
        self.scale = "9c"
        self.model[22].bias_init(640)

    def features(self, x):
        m = self.model
        x1 = m[1](m[0](x))
        x2 = m[2](x1)
        x4 = m[4](m[3](x2))
        x6 = m[6](m[5](x4))
        x8 = m[8](m[7](x6))
        x9 = m[9](x8)
        x12 = m[12](torch.cat((m[10](x9), x6), 1))
        x15 = m[15](torch.cat((m[13](x12), x4), 1))
        x18 = m[18](torch.cat((m[16](x15), x12), 1))
        x21 = m[21](torch.cat((m[19](x18), x9), 1))
        return [x15, x18, x21]

    def forward(self, x):
        return self.model[22](self.features(x))

    def forward_raw(self, x):
        return self.model[22].forward_raw(self.features(x))


class DetectionModelV9cCount(nn.Module):
    """yolov9c with the plain Detect head -- parameter-count pin only (25 590 912 at nc = 80)."""

    def __init__(self, nc: int = 80):
        super().__init__()
        seg = SegmentationModelV9c(nc)
        head = seg.model[22]
        self.body = nn.ModuleList(list(seg.model)[:22])
        self.cv2, self.cv3, self.dfl = head.cv2, head.cv3, head.dfl
