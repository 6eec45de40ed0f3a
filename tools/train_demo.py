"""Fine-tune on the generated defect dataset through the reference script's own call shape
(BscanBased/yolo_seg_train.py:12-19) and report the validator's numbers.  Usage: python tools/train_demo.py [scale] [epochs] [imgsz]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_train_api_gpu import make_defect_dataset  # noqa: E402
from ultralytics import YOLO  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "s"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
imgsz = int(sys.argv[3]) if len(sys.argv) > 3 else 320
tmp = tempfile.mkdtemp()
data = make_defect_dataset(os.path.join(tmp, "data-seg"), n_train=192, n_val=48, size=imgsz)
model = YOLO(f"yolov8{scale}-seg.yaml")
t0 = time.time()
res = model.train(data=data, epochs=epochs, imgsz=imgsz, batch=16, project=os.path.join(tmp, "runs"), name="defect_seg", device=0)
print(f"wall {time.time() - t0:.1f}s  steps {res.optimizer_steps} skipped {res.skipped_steps}")
print({k: round(v, 4) for k, v in res.results_dict.items()})
