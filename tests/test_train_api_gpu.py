"""Training path end to end on the GPU: fused optimizer kernels against torch.optim,
``YOLO.train`` on a generated polygon-label dataset (run directory contract, loss goes down, weights reload
and detect the synthetic defects), ``YOLO.val``."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("kind", ["adamw", "sgd"])
def test_optimizer_kernels_match_torch(kind):
    from defectdetection_viaobjectdetection_amd._capi import check, lib
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    n = 100_003
    p0 = torch.randn(n, generator=g)
    group = (torch.rand(n, generator=g) * 3).to(torch.uint8).clamp_(0, 2)
    ref_params = [torch.nn.Parameter(p0[group == k].clone().to(dev)) for k in range(3)]
    lr, wd = 0.01, 0.05
    if kind == "adamw":
        opt = torch.optim.AdamW([{"params": [ref_params[0]], "weight_decay": wd}, {"params": [ref_params[1]], "weight_decay": 0.0},
                                 {"params": [ref_params[2]], "weight_decay": 0.0, "lr": lr * 0.5}], lr=lr, betas=(0.9, 0.999), eps=1e-8)
    else:
        opt = torch.optim.SGD([{"params": [ref_params[0]], "weight_decay": wd}, {"params": [ref_params[1]], "weight_decay": 0.0},
                               {"params": [ref_params[2]], "weight_decay": 0.0, "lr": lr * 0.5}], lr=lr, momentum=0.9, nesterov=True)
    p = p0.clone().to(dev)
    s1, s2 = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    ema = p.clone()
    ema_ref = p0.clone().to(dev)
    gd = group.to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scale, d = 64.0, 0.9
    for step in range(1, 6):
        grad = torch.randn(n, generator=g).to(dev)
        for k in range(3):
            ref_params[k].grad = grad[gd == k].clone()
        opt.step()
        gs = grad * scale
        if kind == "adamw":
            check(lib.m355_adamw_step(_ptr(p), _ptr(gs), _ptr(s1), _ptr(s2), _ptr(ema), _ptr(gd), n, lr, lr * 0.5, 0.9, 0.999,
                                      1e-8, wd, step, 1.0 / scale, d, st))
        else:
            check(lib.m355_sgd_step(_ptr(p), _ptr(gs), _ptr(s1), _ptr(ema), _ptr(gd), n, lr, lr * 0.5, 0.9, 1, wd, 1.0 / scale, d, st))
        ref = torch.empty(n, device=dev)
        for k in range(3):
            ref[gd == k] = ref_params[k].data
        ema_ref = d * ema_ref + (1 - d) * ref
        torch.testing.assert_close(p, ref, rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(ema, ema_ref, rtol=2e-5, atol=2e-6)


def test_grad_sumsq_and_nonfinite_count():
    from defectdetection_viaobjectdetection_amd._capi import check, lib
    dev = torch.device("cuda", 0)
    x = torch.randn(1_000_001, device=dev)
    out = torch.zeros(int(lib.m355_grad_sumsq_workspace_floats()), device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.m355_grad_sumsq(_ptr(x), x.numel(), _ptr(out), st))
    ss, bad = out[:2].tolist()
    assert bad == 0 and ss == pytest.approx(float((x.double() ** 2).sum()), rel=1e-4)
    check(lib.m355_grad_sumsq(_ptr(x), x.numel(), _ptr(out), st))      # fixed reduction order: the same bits every time
    assert out[:2].tolist() == [ss, 0.0]
    x[5] = float("inf"); x[77] = float("nan"); x[-1] = float("-inf")
    check(lib.m355_grad_sumsq(_ptr(x), x.numel(), _ptr(out), st))
    assert out[:2].tolist()[1] == 3


def make_defect_dataset(root, n_train=24, n_val=8, size=160, seed=0):
    """Noise B-scan stand-ins with 1-2 bright blobs each (rectangles / hexagons) + polygon labels + data yaml."""
    import yaml
    from PIL import Image, ImageDraw
    rng = np.random.default_rng(seed)
    for split, n in (("train", n_train), ("val", n_val)):
        os.makedirs(os.path.join(root, "images", split), exist_ok=True)
        os.makedirs(os.path.join(root, "labels", split), exist_ok=True)
        for i in range(n):
            bg = rng.normal(60, 12, (size, size)).clip(0, 255).astype(np.uint8)
            im = Image.fromarray(bg).convert("RGB")
            dr = ImageDraw.Draw(im)
            rows = []
            for _ in range(int(rng.integers(1, 3))):
                w, h = rng.integers(size // 6, size // 3, 2)
                cx = rng.integers(w // 2 + 2, size - w // 2 - 2)
                cy = rng.integers(h // 2 + 2, size - h // 2 - 2)
                if rng.random() < 0.5:
                    pts = [(cx - w / 2, cy - h / 2), (cx + w / 2, cy - h / 2), (cx + w / 2, cy + h / 2), (cx - w / 2, cy + h / 2)]
                else:
                    pts = [(cx + w / 2 * math.cos(t), cy + h / 2 * math.sin(t)) for t in np.linspace(0, 2 * math.pi, 7)[:-1]]
                dr.polygon(pts, fill=(230, 200, 40))
                rows.append("0 " + " ".join(f"{x / size:.6f} {y / size:.6f}" for x, y in pts))
            im.save(os.path.join(root, "images", split, f"bscan_{i:03d}.png"))
            with open(os.path.join(root, "labels", split, f"bscan_{i:03d}.txt"), "w") as f:
                f.write("\n".join(rows) + "\n")
    ypath = os.path.join(root, "data-seg.yaml")
    with open(ypath, "w") as f:
        yaml.safe_dump({"train": "images/train", "val": "images/val", "names": {0: "defect"}}, f)
    return ypath


def test_train_api_end_to_end(tmp_path):
    from ultralytics import YOLO                      # the shim: the reference script's import line
    data = make_defect_dataset(str(tmp_path / "data-seg"))
    model = YOLO("yolov8n-seg.yaml")
    res = model.train(data=data, epochs=12, imgsz=160, batch=8, project=str(tmp_path / "runs"), name="defect_seg", device=0,
                      warmup_epochs=1.0, verbose=False)
    run = str(tmp_path / "runs" / "defect_seg")
    assert res.save_dir == run
    for f in ("weights/last.pt", "weights/best.pt", "results.csv"):
        assert os.path.isfile(os.path.join(run, f)), f
    hist = res.history
    assert len(hist) == 12
    first = sum(hist[0][k] for k in ("train/box_loss", "train/seg_loss", "train/cls_loss", "train/dfl_loss"))
    last = sum(hist[-1][k] for k in ("train/box_loss", "train/seg_loss", "train/cls_loss", "train/dfl_loss"))
    assert all(math.isfinite(v) for h in hist for v in h.values() if isinstance(v, float))
    assert last < 0.8 * first, (first, last)
    assert res.optimizer_steps > 0 and res.skipped_steps <= res.optimizer_steps
    assert model.nc == 1 and model.names == {0: "defect"}
    # reload the written checkpoint exactly like yolo8_seg_predict.py does and run predict on a val image
    again = YOLO(os.path.join(run, "weights", "best.pt"))
    img = os.path.join(str(tmp_path / "data-seg"), "images", "val", "bscan_000.png")
    r = again.predict(source=img, imgsz=160, conf=0.05, verbose=False)
    assert len(r) == 1 and r[0].boxes.data.shape[1] == 6
    m = again.val(data=data, imgsz=160)
    assert 0.0 <= m.box.map50 <= 1.0 and 0.0 <= m.seg.map50 <= 1.0
    # the run's last validation re-done from the file it saved: last.pt holds the EMA weights the trainer's validator saw
    # after the final epoch (fp32), the kernels are deterministic and batch-invariant -> the same numbers, not a band
    m_last = YOLO(os.path.join(run, "weights", "last.pt")).val(data=data, imgsz=160)
    for key in ("metrics/mAP50(B)", "metrics/mAP50-95(B)", "metrics/mAP50(M)", "metrics/mAP50-95(M)"):
        assert m_last.results_dict[key] == pytest.approx(res.history[-1][key], abs=1e-9), key
    # best.pt is the epoch with the best fitness: its re-validation reproduces that epoch's row
    best_row = max(res.history, key=lambda h: h["fitness"])
    assert m.results_dict["metrics/mAP50(B)"] == pytest.approx(best_row["metrics/mAP50(B)"], abs=1e-9)


def test_train_rejects_unknown_kwargs_and_missing_data(tmp_path):
    from ultralytics import YOLO
    model = YOLO("yolov8n-seg.yaml")
    with pytest.raises(TypeError):
        model.train(data="x.yaml", epochs=1, not_a_real_option=3)
    with pytest.raises(FileNotFoundError):
        model.train(data=str(tmp_path / "missing.yaml"), epochs=1, imgsz=160, batch=2)


def test_two_rank_training_rehearsal_on_one_gpu(tmp_path):
    """The N > 1 training path end to end: two processes launched by torch.distributed.run share this box's one GPU
    (gloo transport, since RCCL refuses two ranks per device); each takes half of every global batch, gradients are
    summed in buckets under backward, rank 0 validates and writes the run directory."""
    import subprocess
    import sys
    data = make_defect_dataset(str(tmp_path / "data-seg"), n_train=16, n_val=4)
    run = tmp_path / "runs"
    script = tmp_path / "ddp_train.py"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script.write_text(
        f"import sys\nsys.path.insert(0, {root!r})\nfrom ultralytics import YOLO\n"
        f"m = YOLO('yolov8n-seg.yaml')\n"
        f"r = m.train(data={data!r}, epochs=3, imgsz=160, batch=8, project={str(run)!r}, name='ddp', device=0, warmup_epochs=1.0, verbose=False, bucket_mb=1)\n"
        f"import os\nprint('RANK', os.environ['RANK'], 'steps', r.optimizer_steps, 'loss', sum(r.history[-1][k] for k in r.history[-1] if k.startswith('train/')) if r.history else -1)\n")
    env = dict(os.environ, M355_DIST_BACKEND="gloo", M355_DIST_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29533", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "RANK 0" in out.stdout and "RANK 1" in out.stdout
    for f in ("weights/last.pt", "weights/best.pt", "results.csv"):
        assert os.path.isfile(os.path.join(str(run), "ddp", f)), f


def test_resume_and_early_stopping(tmp_path):
    from ultralytics import YOLO
    data = make_defect_dataset(str(tmp_path / "data-seg"), n_train=16, n_val=4)
    model = YOLO("yolov8n-seg.yaml")
    r1 = model.train(data=data, epochs=5, imgsz=160, batch=8, project=str(tmp_path / "runs"), name="r", device=0,
                     warmup_epochs=1.0, verbose=False, max_steps=4)        # 2 it/epoch: interrupted after epoch 2
    run = str(tmp_path / "runs" / "r")
    assert len(r1.history) == 2
    again = YOLO(os.path.join(run, "weights", "last.pt"))
    r2 = again.train(resume=True, verbose=False)
    assert r2.save_dir == run and [h["epoch"] for h in r2.history] == [1, 2, 3, 4, 5]
    assert r2.optimizer_steps > r1.optimizer_steps
    # patience: fitness cannot improve on an unlearnable 1-epoch budget with patience 1 -> stops long before 50 epochs
    m3 = YOLO("yolov8n-seg.yaml")
    r3 = m3.train(data=data, epochs=50, imgsz=160, batch=8, project=str(tmp_path / "runs"), name="p", device=0,
                  warmup_epochs=1.0, verbose=False, patience=2, lr0=0.0, optimizer="SGD")
    assert len(r3.history) <= 6


def test_bench_starts_its_own_launcher_for_two_ranks():
    """`python bench.py --gpus 2 ...` exactly as a driver would type it, with no launcher environment: bench.py starts
    torch.distributed.run as a child (before any GPU call), two ranks share this box's one GPU over gloo, rank 0 prints
    ONE JSON line carrying the inference value and the data-parallel training step with its exchange figures."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(M355_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["value"] > 0
    assert line["config"]["global_batch"] == 64 and line["scaling"] == "weak"
    ts = line["train_step"]
    assert "error" not in ts, ts
    for key in ("allreduce_ms", "allreduce_exposed_ms", "overlap_frac", "buckets", "backward_no_exchange_ms", "gradient_mbytes"):
        assert key in ts, key
    assert ts["buckets"] >= 2 and 0.0 <= ts["overlap_frac"] <= 1.0 and math.isfinite(ts["loss"])
