"""Python host wrapper of the libmi355yolo engine (PyTorch-ROCm tensors as device memory / streams).

Stands where ``ultralytics.nn.tasks.SegmentationModel`` + ``SegmentationPredictor.postprocess`` stand
upstream (call site /root/reference/BscanBased/yolo8_seg_predict.py:8): the
arithmetic is in the HIP kernels behind the C-ABI, this file only owns buffers and argument plumbing.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _capi
from ._capi import ConvInfo, ModelDesc, OpInfo, check, lib
from .spec import V9C, ConvSpec, conv_specs, fold_bn


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class SegEngine:
    """One engine per device.  Not thread-safe (one handle, one caller)."""

    def __init__(self, scale: str = "s", nc: int = 1, imgsz: Tuple[int, int] = (640, 640),
                 max_batch: int = 32, device: int = 0, keep_raw: bool = True):
        """keep_raw: also write the raw head maps (`raw_head()`); the predict path and bench.py pass False — the head
        output convs decode their rows in their own epilogue and the raw maps are a parity / debugging output."""
        if not torch.cuda.is_available():
            raise RuntimeError("libmi355yolo needs a gfx950 GPU; there is no CPU fallback")
        self.device = torch.device("cuda", device)
        self.scale, self.nc, self.imgsz, self.max_batch = scale, nc, tuple(imgsz), max_batch
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.init()
            desc = ModelDesc(ord("c" if scale == V9C else scale), nc, imgsz[0], imgsz[1], max_batch)   # 'c': yolov9c-seg
            check(lib.m355_create(C.byref(desc), C.byref(self._h)))
            check(lib.m355_set_keep_raw(self._h, int(keep_raw)), self._h)
        self.num_anchors = lib.m355_num_anchors(self._h)
        self.pred_width = lib.m355_pred_width(self._h)
        ph, pw = C.c_int(), C.c_int()
        check(lib.m355_proto_hw(self._h, C.byref(ph), C.byref(pw)), self._h)
        self.proto_hw = (ph.value, pw.value)
        self.nm = 32
        self.flops_per_image = lib.m355_flops_per_image(self._h)
        self.workspace_bytes = lib.m355_workspace_bytes(self._h)
        self.specs: List[ConvSpec] = conv_specs(scale, nc)
        self._check_graph()

    @staticmethod
    def proto_is_composed(scale: str) -> bool:
        """True when the engine runs Proto's ConvTranspose + 3x3 conv as four composed 2x2 phase convolutions
        (engine.hip build_graph: prototype width a multiple of 64 and M355_NO_PROTOFUSE unset)."""
        import math
        import os
        from .spec import SCALES
        if scale == V9C:
            return "M355_NO_PROTOFUSE" not in os.environ      # 256 prototype channels
        _, width, maxc = SCALES[scale]
        npr = int(math.ceil(min(256, maxc) * width / 8) * 8)
        return npr % 64 == 0 and "M355_NO_PROTOFUSE" not in os.environ

    # ------------------------------------------------------------------ graph / weights
    def conv_infos(self) -> List[ConvInfo]:
        out = []
        for i in range(lib.m355_num_convs(self._h)):
            ci = ConvInfo()
            check(lib.m355_get_conv_info(self._h, i, C.byref(ci)), self._h)
            out.append(ci)
        return out

    def _check_graph(self) -> None:
        infos = self.conv_infos()
        if len(infos) != len(self.specs):
            raise RuntimeError(f"engine reports {len(infos)} convs, host spec has {len(self.specs)}")
        for ci, s in zip(infos, self.specs):
            got = (ci.name.decode(), ci.cin, ci.cout, ci.k, ci.stride, bool(ci.has_bn), bool(ci.transposed))
            want = (s.name, s.cin, s.cout, s.k, s.stride, s.has_bn, s.transposed)
            if got != want:
                raise RuntimeError(f"engine/host graph mismatch: {got} vs {want}")

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Fold BN on the host (A4) and hand fp32 weights to the engine, which packs them to fp16."""
        with torch.cuda.device(self.device):
            for i, s in enumerate(self.specs):
                w, b = fold_bn(sd, s)
                if tuple(w.shape) != s.weight_shape or b.numel() != s.cout:
                    raise ValueError(f"{s.name}: weight shape {tuple(w.shape)} != {s.weight_shape}")
                check(lib.m355_set_conv_weights(self._h, i, _ptr(w), _ptr(b)), self._h)

    # ------------------------------------------------------------------ forward / postprocess
    def forward(self, images_u8_nhwc: torch.Tensor):
        """images: uint8 (B,H,W,3) on this device.  Returns preds f32 (B,A,4+nc+32), protos f16 (B,h,w,32)."""
        x = images_u8_nhwc
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3 or tuple(x.shape[1:3]) != self.imgsz:
            raise ValueError(f"expected uint8 (B,{self.imgsz[0]},{self.imgsz[1]},3), got {x.dtype} {tuple(x.shape)}")
        if not x.is_cuda or not x.is_contiguous():
            raise ValueError("input must be a contiguous CUDA tensor")
        B = x.shape[0]
        preds = torch.empty((B, self.num_anchors, self.pred_width), dtype=torch.float32, device=x.device)
        protos = torch.empty((B, self.proto_hw[0], self.proto_hw[1], self.nm), dtype=torch.float16, device=x.device)
        check(lib.m355_forward(self._h, _ptr(x), B, _ptr(preds), _ptr(protos), _stream()), self._h)
        return preds, protos

    def raw_head(self, batch: int) -> torch.Tensor:
        """Copy of the raw head maps (B,A,64+nc+32) f32 of the last forward (A13 layout)."""
        p, w = C.c_void_p(), C.c_int()
        check(lib.m355_get_raw_head(self._h, C.byref(p), C.byref(w)), self._h)
        out = torch.empty((batch, self.num_anchors, w.value), dtype=torch.float32, device=self.device)
        check(lib.m355_copy_raw_head(self._h, batch, _ptr(out), _stream()), self._h)
        return out

    def postprocess(self, preds: torch.Tensor, protos: Optional[torch.Tensor], conf: float = 0.25,
                    iou: float = 0.7, max_det: int = 300, masks: bool = True, multi_label: bool = False, max_nms: int = 30000):
        """Batched NMS + mask assembly.  Returns dets f32 (B,max_det,38), counts i32 (B),
        masks u8 (B,max_det,H,W) or None.  Only rows < counts[b] are defined.
        ``multi_label`` (upstream's validator mode, nc > 1): every (anchor, class) pair above ``conf`` is a candidate."""
        if multi_label and self.nc > 1:
            return self._postprocess_multilabel(preds, protos, conf, iou, max_det, masks, max_nms)
        B = preds.shape[0]
        dets = torch.empty((B, max_det, 6 + self.nm), dtype=torch.float32, device=preds.device)
        counts = torch.empty((B,), dtype=torch.int32, device=preds.device)
        m = None
        if masks:
            m = torch.empty((B, max_det, self.imgsz[0], self.imgsz[1]), dtype=torch.uint8, device=preds.device)
        check(lib.m355_postprocess(self._h, _ptr(preds), _ptr(protos), B, conf, iou, max_det, _ptr(dets),
                                   _ptr(counts), _ptr(m), _stream()), self._h)
        return dets, counts, m

    def _postprocess_multilabel(self, preds, protos, conf, iou, max_det, masks, max_nms):
        """upstream's ``non_max_suppression(multi_label=True)``: class offsets make the classes independent, so it is one
        single-class NMS launch per class (the kernel and its bit-exact IoU arithmetic unchanged), the per-class survivors
        merged in score order and cut at ``max_det``.  ``max_nms``: scores below an image's max_nms-th largest candidate
        are masked out first (upstream keeps the top ``max_nms`` candidates of an image before the NMS)."""
        B, A, _ = preds.shape
        nc, nm = self.nc, self.nm
        sc = preds[..., 4:4 + nc]
        if A * nc > max_nms:
            kth = sc.reshape(B, A * nc).topk(max_nms, dim=1).values[:, -1]                    # (B,)
            sc = torch.where(sc >= kth[:, None, None], sc, torch.zeros_like(sc))
        all_d, all_valid = [], []
        for c in range(nc):
            pc = torch.cat((preds[..., :4], sc[..., c:c + 1], preds[..., 4 + nc:]), -1).contiguous()
            d = torch.empty((B, max_det, 6 + nm), dtype=torch.float32, device=preds.device)
            n = torch.empty((B,), dtype=torch.int32, device=preds.device)
            check(lib.m355_nms(_ptr(pc), B, A, 1, nm, conf, iou, max_det, _ptr(d), _ptr(n), _stream()))
            d[..., 5] = float(c)
            all_d.append(d)
            all_valid.append(torch.arange(max_det, device=preds.device)[None, :] < n[:, None])
        d = torch.cat(all_d, 1)                                                              # (B, nc * max_det, 6 + nm)
        valid = torch.cat(all_valid, 1)
        key = torch.where(valid, d[..., 4], torch.full_like(d[..., 4], -1.0))
        order = key.argsort(dim=1, descending=True, stable=True)[:, :max_det]
        dets = d.gather(1, order[..., None].expand(B, max_det, 6 + nm)).contiguous()
        counts = valid.sum(1).clamp_max(max_det).to(torch.int32)
        m = None
        if masks:
            m = torch.empty((B, max_det, self.imgsz[0], self.imgsz[1]), dtype=torch.uint8, device=preds.device)
            check(lib.m355_proto_masks(_ptr(dets), _ptr(counts), _ptr(protos), B, max_det, self.proto_hw[0], self.proto_hw[1],
                                       self.imgsz[0], self.imgsz[1], _ptr(m), _stream()))
        return dets, counts, m

    # ------------------------------------------------------------------ measurement hooks
    def op_infos(self) -> List[dict]:
        out = []
        for i in range(lib.m355_num_ops(self._h)):
            oi = OpInfo()
            check(lib.m355_get_op_info(self._h, i, C.byref(oi)), self._h)
            out.append(dict(kernel=oi.kernel.decode(), layer=oi.layer.decode(), flops=oi.flops_per_image,
                            bytes=oi.bytes_per_image, weight_bytes=oi.weight_bytes))
        return out

    def set_profiling(self, enable: bool) -> None:
        check(lib.m355_set_profiling(self._h, int(enable)), self._h)

    def collect_op_times(self):
        """Per-op (sum of milliseconds, launches) since profiling was enabled (HIP events on the launch stream)."""
        n = lib.m355_num_ops(self._h)
        ms = (C.c_double * n)()
        cnt = (C.c_long * n)()
        check(lib.m355_collect_op_times(self._h, ms, cnt), self._h)
        return list(ms), list(cnt)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.m355_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def version() -> str:
    return _capi.version()
