"""Drop-in shim: ``from ultralytics import YOLO`` resolves to the MI355X-native implementation.

The reference scripts (/root/reference/BscanBased/yolo8_seg_predict.py:1, yolo_seg_train.py:1,
yolo/yolo_eval.py:2, yolo/yolo_folder_eval.py:3) import ``YOLO`` from the third-party ``ultralytics``
package.  With the repository root on ``PYTHONPATH`` this package shadows it, so those scripts run
unchanged on the HIP path.  This is NOT the upstream package and contains none of its code.
"""
from defectdetection_viaobjectdetection_amd.model import YOLO  # noqa: F401
from defectdetection_viaobjectdetection_amd.results import Boxes, Masks, Results  # noqa: F401

__version__ = "8.0.0+mi355yolo"
__all__ = ["YOLO", "Results", "Boxes", "Masks"]
