"""Known-answer tests pinning the TRAINING-side oracle: CIoU, DFL, TaskAlignedAssigner,
the segmentation loss, AP / matching, LR and EMA schedules.  CPU only."""
import math

import numpy as np
import pytest
import torch

import yolov8_seg_oracle as orc
import yolov8_seg_train_oracle as tro


def test_ciou_closed_form():
    a = torch.tensor([[0.0, 0.0, 2.0, 2.0]])
    assert float(tro.bbox_iou(a, a)) == pytest.approx(1.0, abs=1e-5)
    b = torch.tensor([[1.0, 0.0, 3.0, 2.0]])                 # same shape, shifted by 1: IoU 1/3, rho^2 = 1, c^2 = 13
    iou = float(tro.bbox_iou(a, b, ciou=False))
    assert iou == pytest.approx(2 / 6, abs=1e-5)
    assert float(tro.bbox_iou(a, b)) == pytest.approx(1 / 3 - 1 / 13, abs=1e-4)
    c = torch.tensor([[10.0, 10.0, 11.0, 14.0]])             # disjoint, different aspect
    ci = float(tro.bbox_iou(a, c))
    v = (4 / math.pi ** 2) * (math.atan(1 / 4) - math.atan(1.0)) ** 2
    alpha = v / (v - 0 + 1)
    rho2 = ((21 - 2) ** 2 + (24 - 2) ** 2) / 4
    assert ci == pytest.approx(0 - (rho2 / (11 ** 2 + 14 ** 2) + v * alpha), abs=1e-4)


def test_dfl_two_hot():
    # target 2.25 -> weights 0.75 on bin 2, 0.25 on bin 3
    logits = torch.zeros(4, 16)
    t = torch.tensor([[2.25, 2.25, 2.25, 2.25]])
    assert float(tro.dfl_loss(logits, t)) == pytest.approx(math.log(16), abs=1e-5)
    logits = torch.full((4, 16), -30.0)
    logits[:, 2] = math.log(0.75)
    logits[:, 3] = math.log(0.25)
    ce = -(0.75 * math.log(0.75) + 0.25 * math.log(0.25))
    assert float(tro.dfl_loss(logits, t)) == pytest.approx(ce, abs=1e-4)
    assert tro.bbox2dist(torch.tensor([[5.0, 5.0]]), torch.tensor([[0.0, 2.0, 30.0, 6.0]]), 15).tolist() == \
        [[5.0, 3.0, pytest.approx(14.99), 1.0]]


def _grid(n=8, stride=8):
    pts, _ = orc.make_anchors([(n, n)], [stride])
    return pts * stride


def test_assigner_topk_inside_and_conflict():
    anc = _grid(8, 8)                                        # 64 anchors at 4,12,...,60
    B, A, nc = 1, anc.shape[0], 2
    gt_boxes = torch.tensor([[[0.0, 0.0, 32.0, 32.0], [16.0, 16.0, 48.0, 48.0]]])
    gt_labels = torch.tensor([[[0.0], [1.0]]])
    mask_gt = torch.ones(1, 2, 1, dtype=torch.bool)
    # every anchor predicts its own 16x16 box centred on the anchor; scores 0.5 for both classes
    pd_boxes = torch.cat((anc - 8, anc + 8), 1)[None]
    pd_scores = torch.full((B, A, nc), 0.5)
    tb, ts, fg, idx = tro.task_aligned_assign(pd_scores, pd_boxes, anc, gt_labels, gt_boxes, mask_gt, topk=4)
    inside0 = ((anc > 0) & (anc < 32)).all(1)
    inside1 = ((anc > 16) & (anc < 48)).all(1)
    assert bool((fg[0] <= (inside0 | inside1)).all())        # positives are centres strictly inside a GT
    assert 4 <= int(fg.sum()) <= 8                            # top-4 per GT, conflicts resolved to one GT
    both = inside0 & inside1
    for a in torch.nonzero(fg[0] & both).flatten().tolist(): # claimed by both -> the GT with the larger CIoU
        i0 = float(tro.bbox_iou(gt_boxes[0, 0], pd_boxes[0, a]).clamp(0))
        i1 = float(tro.bbox_iou(gt_boxes[0, 1], pd_boxes[0, a]).clamp(0))
        assert int(idx[0, a]) == (0 if i0 >= i1 else 1)
    # target scores: one-hot of the assigned class, scaled so that the best anchor of each GT gets its max overlap
    for g in range(2):
        sel = fg[0] & (idx[0] == g)
        assert bool((ts[0, sel, 1 - g] == 0).all()) and float(ts[0, sel, g].max()) > 0
    assert bool((ts[0, ~fg[0]] == 0).all())
    # no GT -> nothing assigned
    tb, ts, fg, idx = tro.task_aligned_assign(pd_scores, pd_boxes, anc, gt_labels[:, :0], gt_boxes[:, :0], mask_gt[:, :0])
    assert not fg.any() and float(ts.sum()) == 0


def _batch():
    masks = torch.zeros(2, 160, 160)
    masks[0, 20:60, 10:70] = 1
    masks[0, 100:120, 100:140] = 2
    return {"batch_idx": torch.tensor([0, 0]), "cls": torch.tensor([0, 0]),
            "bboxes": torch.tensor([[0.25, 0.25, 0.375, 0.25], [0.75, 0.6875, 0.25, 0.125]]), "masks": masks}


def test_segmentation_loss_runs_and_backprops():
    torch.manual_seed(0)
    model = orc.SegmentationModel("n", 1).train()
    x = torch.rand(2, 3, 640, 640)
    raw, mc, proto = model.forward_raw(x)
    loss, items = tro.segmentation_loss(raw, mc, proto, _batch(), 1, (640, 640))
    assert loss.ndim == 0 and torch.isfinite(loss) and items.shape == (4,) and bool((items >= 0).all())
    loss.backward()
    g = model.model[0].conv.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    # image 1 has no labels: its class logits only see background targets (gradient pushes them down)
    assert model.model[22].cv3[0][2].bias.grad is not None


def test_loss_empty_batch_keeps_graph():
    model = orc.SegmentationModel("n", 1).train()
    raw, mc, proto = model.forward_raw(torch.rand(1, 3, 320, 320))
    empty = {"batch_idx": torch.zeros(0), "cls": torch.zeros(0), "bboxes": torch.zeros(0, 4), "masks": torch.zeros(1, 80, 80)}
    loss, items = tro.segmentation_loss(raw, mc, proto, empty, 1, (320, 320))
    assert float(items[0]) == 0 and float(items[1]) == 0 and float(items[3]) == 0 and float(items[2]) > 0
    loss.backward()


def test_ap_known_answers():
    # perfect detector
    ap, cls = tro.ap_per_class(np.ones((3, 1), bool), np.array([0.9, 0.8, 0.7]), np.zeros(3), np.zeros(3))
    assert ap[0, 0] == pytest.approx(0.995, abs=2e-3)      # 101-point interpolation: a perfect detector scores 0.995
    # TP, FP, TP with 2 GT: PR points (0.5,1), (0.5,0.5), (1,2/3) -> envelope 1 until r=.5 then 2/3
    tp = np.array([[1], [0], [1]], bool)
    ap, _ = tro.ap_per_class(tp, np.array([0.9, 0.8, 0.7]), np.zeros(3), np.zeros(2))
    assert ap[0, 0] == pytest.approx(0.5 * 1 + 0.5 * (2 / 3), abs=0.02)


def test_match_predictions_greedy_unique():
    iou = np.array([[0.9, 0.6, 0.0], [0.55, 0.8, 0.3]])                 # 2 GT x 3 preds
    c = tro.match_predictions(np.zeros(3), np.zeros(2), iou)
    assert c[:, 0].tolist() == [True, True, False]                      # thr 0.5: p0<->g0, p1<->g1
    assert c[:, 6].tolist() == [True, True, False]                      # thr 0.8
    assert c[:, 8].tolist() == [True, False, False]                     # thr 0.9 (>= is inclusive)
    c = tro.match_predictions(np.array([1, 0, 0]), np.zeros(2), iou)    # class mismatch for p0
    assert c[:, 0].tolist() == [False, True, False]


def test_schedules():
    assert tro.lr_lambda(0, 30) == pytest.approx(1.0) and tro.lr_lambda(30, 30) == pytest.approx(0.01)
    assert tro.lr_lambda(15, 30) == pytest.approx(0.505)
    assert tro.ema_decay(0) == 0 and tro.ema_decay(2000) == pytest.approx(0.9999 * (1 - math.exp(-1)))
