"""MI355X-native YOLOv8-seg hot path for PAUT B-scan defect detection.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) over the C-ABI of
``libmi355yolo.so`` (hand-written HIP kernels for gfx950).  See DESIGN.md / INTEGRATION.md.
There is no CPU fallback: importing ``engine`` (or anything that computes) requires the built library.
"""
__version__ = "0.1.0"

__all__ = ["__version__"]
