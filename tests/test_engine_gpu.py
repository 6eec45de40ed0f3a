"""End-to-end parity of the HIP engine (through the C-ABI) against the CPU oracle on seeded inputs.

Stated tolerances (SURVEY 8d / BASELINE.md section 4).  Two references:
  * the fp32 oracle: raw maps rel-L2 <= 1e-2; scores <= 2e-3 abs and boxes <= 0.5 px for 99 % of the anchors;
  * the MAXIMA over all anchors are bounded by what the number format itself costs on the same inputs: the oracle in the
    engine's number format (fp16 storage at the engine's rounding points, fp32 sums; oracle/engine_format_oracle.py)
    against the fp32 oracle, measured in the test (maxima x 2, rms x 1.25).  (A 60-layer fp16-storage network is chaotic at the ulp level:
    two correct implementations of the same format are as far from each other as each is from fp32 -- measured)
  NMS keep-set / order          exact when both sides consume the same preds (bit-exact IoU arithmetic)
  masks                         >= 99.5 % pixel agreement
"""
import numpy as np
import pytest
import torch

from helpers import build_oracle, synthetic_bscans

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def setup_s(cuda_device):
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", 1, seed=0)
    eng = SegEngine("s", 1, (640, 640), max_batch=4)
    eng.load_state_dict(sd)
    oracle = build_oracle("s", 1, sd)
    imgs = synthetic_bscans(2, seed=1)
    x = torch.from_numpy(imgs.transpose(0, 3, 1, 2).copy()).float() / 255.0
    import engine_format_oracle as efo
    fmt = efo.to_engine_format(build_oracle("s", 1, sd))
    with torch.no_grad():
        raw, mc, protos = oracle.forward_raw(x)
        preds, _ = oracle(x)
        f_raw, f_mc, f_protos = fmt.forward_raw(x)
        f_preds, _ = fmt(x)
    return dict(eng=eng, oracle=oracle, imgs=imgs, raw=raw, mc=mc, protos=protos, preds=preds, sd=sd,
                f_raw=f_raw, f_mc=f_mc, f_protos=f_protos, f_preds=f_preds)


def test_graph_matches_spec(setup_s):
    eng = setup_s["eng"]
    assert eng.num_anchors == 8400 and eng.pred_width == 37 and eng.proto_hw == (160, 160)
    # conv-only FLOPs per image, SURVEY 8d: 39.92 GFLOP for s-seg nc=1
    assert abs(eng.flops_per_image - 2 * 19957606400) < 1


def test_forward_parity(setup_s, cuda_device):
    s = setup_s
    eng = s["eng"]
    d_in = torch.from_numpy(s["imgs"]).to(cuda_device)
    preds, protos = eng.forward(d_in)
    raw = eng.raw_head(2)
    torch.cuda.synchronize()
    B = 2
    g_raw = raw.cpu()
    assert torch.isfinite(g_raw).all()
    g_pred = preds.cpu()
    g_pr = protos.float().cpu().permute(0, 3, 1, 2)

    def against(raw_l, mc, pr, pred):
        o_raw = torch.cat([r.view(B, 65, -1) for r in raw_l], 2)                 # (B,65,A)
        o_raw = torch.cat((o_raw, mc), 1).permute(0, 2, 1).contiguous()          # (B,A,97)
        o_pred = pred.permute(0, 2, 1)
        return dict(box=rel_l2(g_raw[..., :64], o_raw[..., :64]), cls=float((g_raw[..., 64] - o_raw[..., 64]).abs().max()),
                    mc=rel_l2(g_raw[..., 65:], o_raw[..., 65:]), pr=rel_l2(g_pr, pr),
                    dbox=float((g_pred[..., :4] - o_pred[..., :4]).abs().max()),
                    dbox_med=float((g_pred[..., :4] - o_pred[..., :4]).abs().median()),
                    dscore=float((g_pred[..., 4] - o_pred[..., 4]).abs().max()))
    vf = against(s["f_raw"], s["f_mc"], s["f_protos"], s["f_preds"])             # HIP vs the engine-format oracle
    vo = against(s["raw"], s["mc"], s["protos"], s["preds"])                     # HIP vs the fp32 oracle
    # the number format's own cost on these inputs: format oracle vs fp32 oracle
    f_cls = torch.cat([r.view(B, 65, -1) for r in s["f_raw"]], 2)[:, 64]
    o_cls = torch.cat([r.view(B, 65, -1) for r in s["raw"]], 2)[:, 64]
    fo = dict(cls=float((f_cls - o_cls).abs().max()), cls_rms=float((f_cls - o_cls).pow(2).mean().sqrt()),
              dbox=float((s["f_preds"][:, :4] - s["preds"][:, :4]).abs().max()),
              dscore=float((s["f_preds"][:, 4] - s["preds"][:, 4]).abs().max()))
    g_cls_rms = float((g_raw[..., 64] - o_cls).pow(2).mean().sqrt())
    o_pred = s["preds"].permute(0, 2, 1)
    q = lambda t, f: float(t.flatten().kthvalue(max(1, int(t.numel() * f)))[0])
    dsc, dbx = (g_pred[..., 4] - o_pred[..., 4]).abs(), (g_pred[..., :4] - o_pred[..., :4]).abs()
    for name, v in (("format oracle", vf), ("fp32 oracle", vo)):
        print(f"HIP vs {name}: raw box rel-L2 {v['box']:.3e} coef {v['mc']:.3e} proto {v['pr']:.3e} | class logit max {v['cls']:.3e} | "
              f"box px median {v['dbox_med']:.5f} max {v['dbox']:.4f} | score max {v['dscore']:.3e}")
    print(f"HIP vs fp32 oracle quantiles: score p99 {q(dsc, .99):.2e} p99.9 {q(dsc, .999):.2e}; box px p99 {q(dbx, .99):.3f} p99.9 {q(dbx, .999):.3f}; "
          f"class logit rms {g_cls_rms:.2e}")
    print(f"format floor (format oracle vs fp32 oracle): class logit max {fo['cls']:.3e} rms {fo['cls_rms']:.2e} box max {fo['dbox']:.4f} px "
          f"score max {fo['dscore']:.3e}")
    # whole-net bounds as stated: rel-L2 of the raw maps, and the stated score / box tolerances for all but
    # the heaviest 0.1 % / 1 % of the 2 x 8400 anchors
    assert vo["box"] <= 1e-2 and vo["mc"] <= 1e-2 and vo["pr"] <= 1e-2
    assert q(dsc, .99) <= 2e-3 and q(dbx, .99) <= 0.5
    # the maxima: a 60-layer fp16-storage network is chaotic at the ulp level, so the bound is what
    # the number format itself costs on these inputs -- an independent CPU implementation in the same format, measured here
    # (maxima of a heavy-tailed noise: x 1.5 -- measured 0.99 x (score), 0.90 x (box), 1.03 x (class logit); the rms, a stable statistic: x 1.25)
    assert vo["dscore"] <= 1.5 * fo["dscore"] and vo["dbox"] <= 1.5 * fo["dbox"] and vo["cls"] <= 1.5 * fo["cls"]
    assert g_cls_rms <= 1.25 * fo["cls_rms"] + 1e-4
    # two implementations of the same format are as far from each other as each is from fp32, not closer
    assert vf["dscore"] <= 1.5 * fo["dscore"] and vf["dbox"] <= 1.5 * fo["dbox"]


def test_decode_parity_same_raw(setup_s, cuda_device):
    """Decode kernel alone: feed the ORACLE's raw maps -> preds must match the oracle decode tightly."""
    import ctypes as C
    from defectdetection_viaobjectdetection_amd import _capi
    s = setup_s
    B = 2
    o_raw = torch.cat([r.view(B, 65, -1) for r in s["raw"]], 2)
    o_raw = torch.cat((o_raw, s["mc"]), 1).permute(0, 2, 1).contiguous()
    d_raw = o_raw.to(cuda_device)
    d_pred = torch.empty((B, 8400, 37), dtype=torch.float32, device=cuda_device)
    _capi.check(_capi.lib.m355_head_decode(C.c_void_p(d_raw.data_ptr()), B, 640, 640, 1, C.c_void_p(d_pred.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    o_pred = s["preds"].permute(0, 2, 1)
    g = d_pred.cpu()
    assert float((g[..., :4] - o_pred[..., :4]).abs().max()) <= 1e-2   # pixels
    assert float((g[..., 4] - o_pred[..., 4]).abs().max()) <= 1e-6
    assert torch.equal(g[..., 5:], o_pred[..., 5:])


def test_nms_exact_same_preds(setup_s, cuda_device):
    """NMS keep-set, order and rows are bit-exact when both sides consume the same preds."""
    import yolov8_seg_oracle as orc
    s = setup_s
    eng = s["eng"]
    # GPU preds are the common input (they contain a realistic number of candidates)
    d_in = torch.from_numpy(s["imgs"]).to(cuda_device)
    preds, protos = eng.forward(d_in)
    for conf, iou, max_det in ((0.25, 0.7, 300), (0.001, 0.7, 300), (0.05, 0.45, 50)):
        dets, counts, _ = eng.postprocess(preds, protos, conf, iou, max_det, masks=False)
        torch.cuda.synchronize()
        ref = orc.non_max_suppression(preds.cpu().permute(0, 2, 1).numpy(), 1, conf, iou, max_det)
        for b in range(preds.shape[0]):
            n = int(counts[b])
            assert n == ref[b].shape[0], (conf, iou, n, ref[b].shape)
            got = dets[b, :n].cpu().numpy()
            assert np.array_equal(got, ref[b]), (conf, iou)
        print(f"conf {conf} iou {iou}: kept {[int(c) for c in counts]}")


def test_masks_parity(setup_s, cuda_device):
    import yolov8_seg_oracle as orc
    s = setup_s
    eng = s["eng"]
    d_in = torch.from_numpy(s["imgs"]).to(cuda_device)
    preds, protos = eng.forward(d_in)
    dets, counts, masks = eng.postprocess(preds, protos, 0.25, 0.7, 300, masks=True)
    torch.cuda.synchronize()
    total = agree = 0
    for b in range(preds.shape[0]):
        n = int(counts[b])
        if n == 0:
            continue
        d = dets[b, :n].cpu()
        ref = orc.process_mask(protos[b].float().cpu().permute(2, 0, 1), d[:, 6:], d[:, :4], (640, 640))
        got = masks[b, :n].cpu().bool()
        total += ref.numel()
        agree += int((got == ref).sum())
    assert total > 0, "synthetic weights must produce detections"
    frac = agree / total
    print(f"mask pixel agreement {frac:.6f} over {total} px")
    assert frac >= 0.995


def test_upsample_read_through_equals_materialised_upsample(cuda_device):
    """model.12.cv1 / model.15.cv1 read the upsampled half of their input through the gather (no upsample kernel);
    with M355_NO_UPFUSE the upsample kernel writes it first.  Same bytes in, same bits out."""
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    from helpers import synthetic_bscans
    sd = synthetic_state_dict("s", 1, seed=0)
    imgs = torch.from_numpy(synthetic_bscans(3, seed=11)[:, :320, :384].copy()).to(cuda_device)
    outs = []
    for fuse in (True, False):
        if fuse:
            os.environ.pop("M355_NO_UPFUSE", None)
        else:
            os.environ["M355_NO_UPFUSE"] = "1"
        try:
            eng = SegEngine("s", 1, (320, 384), max_batch=3)
            kinds = [o["kernel"] for o in eng.op_infos()]
            assert ("upsample2x" in kinds) == (not fuse)
            eng.load_state_dict(sd)
            p, q = eng.forward(imgs)
            torch.cuda.synchronize()
            outs.append((p.clone(), q.clone()))
            eng.close()
        finally:
            os.environ.pop("M355_NO_UPFUSE", None)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_composed_proto_matches_two_step_proto(cuda_device):
    """ConvTranspose -> 3x3 conv of the Proto branch runs as four composed 2x2 phase convs; with M355_NO_PROTOFUSE the
    two layers run separately (fp16 intermediate).  Same function, different rounding: prototypes agree to ~1e-3,
    and the composed form is the one closer to the fp32 oracle (checked by the other tests' tolerances)."""
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    from helpers import synthetic_bscans
    _composed_proto_case("n", cuda_device)      # 64 prototype channels: 64x128 tile, cv3 separate
    _composed_proto_case("s", cuda_device)      # 128 channels: 128x128 tile with proto.cv3 in the epilogue


def _composed_proto_case(scale, cuda_device):
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    from helpers import synthetic_bscans
    sd = synthetic_state_dict(scale, 1, seed=2)
    imgs = torch.from_numpy(synthetic_bscans(2, seed=5)[:, :256, :320].copy()).to(cuda_device)
    outs = []
    for fuse in (True, False):
        if fuse:
            os.environ.pop("M355_NO_PROTOFUSE", None)
        else:
            os.environ["M355_NO_PROTOFUSE"] = "1"
        try:
            eng = SegEngine(scale, 1, (256, 320), max_batch=2)
            kinds = [o["kernel"] for o in eng.op_infos()]
            assert any("phase" in k for k in kinds) == fuse
            assert any("phase+1x1" in k for k in kinds) == (fuse and scale == "s")
            eng.load_state_dict(sd)
            p, q = eng.forward(imgs)
            torch.cuda.synchronize()
            outs.append((p.clone(), q.float().clone()))
            eng.close()
        finally:
            os.environ.pop("M355_NO_PROTOFUSE", None)
    assert torch.equal(outs[0][0], outs[1][0])                    # detections do not depend on the proto branch
    err = float((outs[0][1] - outs[1][1]).norm() / outs[1][1].norm())
    assert err <= 3e-3, err
    # borders included: the first / last rows and columns carry the border-class bias
    edge = torch.cat((outs[0][1][:, 0].flatten(), outs[0][1][:, -1].flatten(), outs[0][1][:, :, 0].flatten(), outs[0][1][:, :, -1].flatten()))
    edge_ref = torch.cat((outs[1][1][:, 0].flatten(), outs[1][1][:, -1].flatten(), outs[1][1][:, :, 0].flatten(), outs[1][1][:, :, -1].flatten()))
    assert float((edge - edge_ref).norm() / edge_ref.norm()) <= 3e-3


def test_stream_lanes_equal_single_stream(cuda_device):
    """Proto and the head levels run on engine-owned side streams (plan_lanes in engine.hip); M355_NO_LANES puts every
    op on the caller's stream.  Same kernels, same inputs: bit-identical outputs, also when forwards follow each other
    without a host synchronisation (the join at the end of a forward orders the next one) and when two engines run on
    two caller streams at once (bench.py's two batches in flight)."""
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", 1, seed=0)
    batches = [torch.from_numpy(synthetic_bscans(3, seed=20 + i)[:, :256, :320].copy()).to(cuda_device) for i in range(3)]

    def make(no_lanes):
        if no_lanes:
            os.environ["M355_NO_LANES"] = "1"
        try:
            eng = SegEngine("s", 1, (256, 320), max_batch=3)
        finally:
            os.environ.pop("M355_NO_LANES", None)
        eng.load_state_dict(sd)
        return eng

    ref_eng = make(True)
    ref = []
    for x in batches:
        p, q = ref_eng.forward(x)
        torch.cuda.synchronize()
        ref.append((p.clone(), q.clone()))
    ref_eng.close()

    a, b = make(False), make(False)
    # back-to-back forwards on one stream, results copied out by stream-ordered clones
    got = []
    for x in batches * 2:
        p, q = a.forward(x)
        got.append((p.clone(), q.clone()))
    torch.cuda.synchronize()
    for i, (p, q) in enumerate(got):
        assert torch.equal(p, ref[i % 3][0]) and torch.equal(q, ref[i % 3][1]), i
    # two engines, two streams, concurrently
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):
        with torch.cuda.stream(s1):
            p1, q1 = a.forward(batches[rep])
            outs.append((rep, p1.clone(), q1.clone()))
        with torch.cuda.stream(s2):
            p2, q2 = b.forward(batches[(rep + 1) % 3])
            outs.append(((rep + 1) % 3, p2.clone(), q2.clone()))
    torch.cuda.synchronize()
    for i, p, q in outs:
        assert torch.equal(p, ref[i][0]) and torch.equal(q, ref[i][1]), i
    a.close()
    b.close()


def test_conv_plus_cv1_fusion_is_bit_identical(cuda_device):
    """model.1 -> model.2.cv1 and model.3 -> model.4.cv1 (s scale: 64 and 128 channels = one channel tile) run as one
    launch each, the 1x1 in the stride-2 conv's epilogue through LDS; M355_NO_CVFUSE runs them separately.  The fp16
    intermediate and the K order of the 1x1 are the same in both forms: identical bits.  At the n scale the 64-channel
    pair is model.3 -> model.4.cv1; its 128-channel pair has too few pixels here for the 128-channel tile."""
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    # the 32 -> 64 pair has its own patch kernel by default (next test) and the separate 128 -> 128 cv1 would run on the
    # weights-in-registers 1x1 kernel (another summation order), a separate stride-2 conv on the row-slab kernel (round 4) likewise:
    # this test is about the im2col kernel's epilogue fusion
    knobs = {"M355_NO_S2C32": "1", "M355_NO_W1": "1", "M355_NO_S2C64": "1", "M355_NO_PLANES_S2": "1"}
    os.environ.update(knobs)
    try:
        _cv1_fusion_cases(cuda_device)
    finally:
        for k in knobs:
            os.environ.pop(k, None)


def _cv1_fusion_cases(cuda_device):
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    for scale, nfused in (("s", 2), ("n", 1)):
        sd = synthetic_state_dict(scale, 1, seed=0)
        # 7 images of 608 x 640: the 76 x 80 map of model.3 gives 332.5 pixel tiles (a partial one), enough for the 128-channel tile
        imgs = torch.from_numpy(synthetic_bscans(7, seed=31)[:, :608, :].copy()).to(cuda_device)
        outs = []
        saved = os.environ.pop("M355_NO_CVFUSE", None)
        for fuse in (True, False):
            if not fuse:
                os.environ["M355_NO_CVFUSE"] = "1"
            try:
                eng = SegEngine(scale, 1, (608, 640), max_batch=7)
            finally:
                os.environ.pop("M355_NO_CVFUSE", None)
            kinds = [o["kernel"] for o in eng.op_infos()]
            assert sum("+1x1>" in k and "phase" not in k for k in kinds) == (nfused if fuse else 0), kinds
            eng.load_state_dict(sd)
            p, q = eng.forward(imgs)
            torch.cuda.synchronize()
            outs.append((p.clone(), q.clone()))
            eng.close()
        if saved is not None:
            os.environ["M355_NO_CVFUSE"] = saved
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), scale


def test_s2c32_patch_kernel_matches_the_im2col_form(cuda_device):
    """model.1 + model.2.cv1 of the s scale on conv3x3_s2c32.hip (32x32x16 MFMA, staged patch, resident weights) against the
    same pair on the im2col kernel with the epilogue fusion: same fp16 intermediate, different summation order.  Every later
    layer is identical in the two engines, so the network outputs may differ by the ulp-level noise a deep fp16 network
    amplifies (~1e-3) and no more; a wrong tap, channel or pixel mapping would be an O(1) difference
    (and fails test_forward_parity against the oracle as well)."""
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", 1, seed=0)
    imgs = torch.from_numpy(synthetic_bscans(5, seed=33)).to(cuda_device)
    outs = []
    for off in (False, True):
        if off:
            os.environ["M355_NO_S2C32"] = "1"
        try:
            eng = SegEngine("s", 1, (640, 640), max_batch=5)
        finally:
            os.environ.pop("M355_NO_S2C32", None)
        kinds = [o["kernel"] for o in eng.op_infos()]
        assert any("s2c32" in k for k in kinds) == (not off), kinds
        eng.load_state_dict(sd)
        p, q = eng.forward(imgs)
        raw = eng.raw_head(5)
        torch.cuda.synchronize()
        outs.append((p.clone(), q.clone(), raw.clone()))
        eng.close()
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())  # noqa: E731
    e_raw, e_pr = rel(outs[0][2], outs[1][2]), rel(outs[0][1], outs[1][1])
    print(f"s2c32 vs im2col form: raw head rel-L2 {e_raw:.2e}, prototypes rel-L2 {e_pr:.2e}")
    assert e_raw <= 2e-3 and e_pr <= 2e-3


@pytest.mark.parametrize("shape,batch", [((640, 640), 5), ((320, 384), 3)])
def test_two_team_stem_launch_is_bit_identical(cuda_device, shape, batch):
    """conv_stem_c2.hip (the default): model.0 + model.1 + model.2.cv1 with team X building the next tile's patch image while
    team Y runs the two convolutions out of registers, against the lockstep launch (conv_stem_s2c32.hip, M355_NO_STEM2=1).  Same rounding points,
    same K orders: identical bits in every network output."""
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", 1, seed=0)
    imgs = torch.from_numpy(synthetic_bscans(batch, seed=61)[:, :shape[0], :shape[1]].copy()).to(cuda_device)
    eng = SegEngine("s", 1, shape, max_batch=batch)
    assert any(k.startswith("stem+conv3x3_s2c32") for k in (o["kernel"] for o in eng.op_infos()))
    eng.load_state_dict(sd)
    outs = []
    for two_team in (False, True):
        if not two_team:
            os.environ["M355_NO_STEM2"] = "1"
        try:
            p, q = eng.forward(imgs)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("M355_NO_STEM2", None)
        outs.append((p.clone(), q.clone()))
    eng.close()
    assert torch.isfinite(outs[1][0]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_sub_batched_leading_ops_are_bit_identical(cuda_device):
    """M355_SUBBATCH=8 runs the large-map ops at the head of the graph over 8 images at a time (an experiment: the tensors
    then fit the Infinity Cache between producer and consumer; measured slower, so it is off by default).  Same
    kernels per image: identical bits, also with a ragged last pass (20 = 8 + 8 + 4)."""
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", 1, seed=0)
    imgs = torch.from_numpy(synthetic_bscans(20, seed=41)[:, :320, :384].copy()).to(cuda_device)
    outs = []
    for sub in (True, False):
        if sub:
            os.environ["M355_SUBBATCH"] = "8"
        try:
            eng = SegEngine("s", 1, (320, 384), max_batch=20)
        finally:
            os.environ.pop("M355_SUBBATCH", None)
        eng.load_state_dict(sd)
        p, q = eng.forward(imgs)
        raw = eng.raw_head(20)
        torch.cuda.synchronize()
        outs.append((p.clone(), q.clone(), raw.clone()))
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_decode_in_the_head_epilogue_is_bit_identical(cuda_device):
    """M355_DECFUSE=1 (an experiment, measured slower, off by default): the three head output convs decode their rows
    themselves (DFL expectation, dist2bbox, sigmoid, coefficient copy in the conv epilogue); the default writes the raw
    maps and launches head_decode_kernel.  Same arithmetic on the same
    fp32 rows: identical predictions, with and without the raw maps, and identical raw maps when they are kept.
    Partial pixel tiles: 7 images of 608 x 640 (76 x 80, 38 x 40 and 19 x 20 maps)."""
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", 1, seed=0)
    imgs = torch.from_numpy(synthetic_bscans(7, seed=51)[:, :608, :].copy()).to(cuda_device)
    outs = {}
    for mode in ("fused", "fused_keep_raw", "separate"):
        if mode != "separate":
            os.environ["M355_DECFUSE"] = "1"
        try:
            eng = SegEngine("s", 1, (608, 640), max_batch=7, keep_raw=(mode != "fused"))
        finally:
            os.environ.pop("M355_DECFUSE", None)
        kinds = [o["kernel"] for o in eng.op_infos()]
        assert sum("+decode" in k for k in kinds) == (0 if mode == "separate" else 3)
        assert ("head_decode" in kinds) == True   # the op stays in the list; it launches nothing when the convs decode
        eng.load_state_dict(sd)
        p, q = eng.forward(imgs)
        raw = eng.raw_head(7) if mode != "fused" else None
        torch.cuda.synchronize()
        outs[mode] = (p.clone(), q.clone(), None if raw is None else raw.clone())
        if mode == "fused":
            with pytest.raises(RuntimeError):
                eng.raw_head(7)
        eng.close()
    assert torch.equal(outs["fused"][0], outs["separate"][0]) and torch.equal(outs["fused_keep_raw"][0], outs["separate"][0])
    assert torch.equal(outs["fused"][1], outs["separate"][1])
    assert torch.equal(outs["fused_keep_raw"][2], outs["separate"][2])


@pytest.mark.parametrize("shape,batch,nc", [((608, 640), 7, 1), ((640, 640), 3, 1), ((320, 320), 5, 1), ((320, 384), 4, 3), ((256, 256), 2, 20)])
def test_head_levels_as_conv_plus_decode_launches(cuda_device, shape, batch, nc):
    """Predict path (raw head maps not kept): each head level's three output convs and the decode of its rows run as ONE
    launch (head_tail.hip) and head_decode_kernel is not launched; with the raw maps kept, the im2col launch per level + the
    decode launch.  Same arithmetic after the dot products (the decode is restated operation for operation); the dot products
    themselves run on another MFMA shape (32x32x16 vs 16x16x32) and still come out identical: the rows are bit-identical.
    Shapes: 608 x 640 (76 x 80 map: fast row stores; 38 x 40 and 19 x 20: pixel blocks that cross image boundaries), three
    images of 640 x 640 (a partial last tile on the 40 x 40 level), 320 x 320 (10 x 10 = 100 anchors per image); nc = 3 and 20:
    class rows and the coefficient offset behind them (row width 4 + nc + 32: 39 floats has no 16-byte row alignment)."""
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", nc, seed=0)
    imgs = torch.from_numpy(synthetic_bscans(batch, seed=52)[:, :shape[0], :shape[1]].copy()).to(cuda_device)
    outs = {}
    for keep in (False, True):
        eng = SegEngine("s", nc, shape, max_batch=batch, keep_raw=keep)
        kinds = [o["kernel"] for o in eng.op_infos()]
        assert sum(k.startswith("head_tail") for k in kinds) == (0 if keep else 3)
        eng.load_state_dict(sd)
        p, q = eng.forward(imgs)
        torch.cuda.synchronize()
        outs[keep] = (p.clone(), q.clone())
        if not keep:
            with pytest.raises(RuntimeError):
                eng.raw_head(batch)
        eng.close()
    a, b = outs[False][0], outs[True][0]
    assert a.shape == b.shape and torch.isfinite(a).all()
    assert torch.equal(outs[False][1], outs[True][1])                       # prototypes: untouched
    d = (a - b).abs()
    print(f"{shape} b{batch} nc{nc}: box max {float(d[..., :4].max()):.2e} px, score max {float(d[..., 4:4 + nc].max()):.2e}, "
          f"coef max {float(d[..., 4 + nc:].max()):.2e}")
    # measured: identical bits on every shape (the MFMA sums of these K = 64 / 128 / 32 dot products come out the same in both
    # shapes of the instruction); asserted as equality so that a change of either side shows
    assert torch.equal(a, b)


def test_head_tail_is_all_levels_or_none(cuda_device):
    """The head_tail launch is chosen for all three head levels of a forward or for none (engine.hip, top of m355_forward): its 31-bit
    byte-offset bound is reached by the stride-8 level first (batch >= 749 at 640 x 640), and a per-level choice would let the decode
    launch overwrite the rows the two smaller levels had written with whatever the raw buffer holds (round-3 advisor finding).
    M355_HEADTAIL_MAXM (testing knob, read at m355_create) makes every level with more pixels than that ineligible: with the
    stride-8 level forced out, the predictions must be those of the im2col + decode path, bit for bit."""
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", 1, seed=0)
    imgs = torch.from_numpy(synthetic_bscans(3, seed=53)[:, :320, :320].copy()).to(cuda_device)
    outs = {}
    for name, keep, maxm in (("reference", True, None), ("forced_out", False, 3 * 20 * 20 + 1), ("head_tail", False, None)):
        if maxm is not None:
            os.environ["M355_HEADTAIL_MAXM"] = str(maxm)      # 3 x 40 x 40 pixels of the stride-8 level exceed it, the other two do not
        try:
            eng = SegEngine("s", 1, (320, 320), max_batch=3, keep_raw=keep)
        finally:
            os.environ.pop("M355_HEADTAIL_MAXM", None)
        eng.load_state_dict(sd)
        eng.forward(imgs)                                      # (a first forward leaves stale rows in the raw buffer)
        p, _ = eng.forward(torch.flip(imgs, (0,)))
        torch.cuda.synchronize()
        outs[name] = p.clone()
        eng.close()
    assert torch.isfinite(outs["forced_out"]).all()
    assert torch.equal(outs["forced_out"], outs["reference"])
    assert torch.equal(outs["head_tail"], outs["reference"])


def test_stem_fused_into_the_patch_kernel_matches_the_two_launches(cuda_device):
    """model.0 + model.1 + model.2.cv1 in one launch (conv_stem_s2c32.hip: the patch of the stem output is computed from the
    uint8 window in LDS and never stored) against stem launch + patch kernel.  Same MFMA shape and tap order for the stem, so
    the only differences are the last-bit rounding of its epilogue; batch 5 covers every border class of the tiles (first /
    last tile rows and columns: the zero padding of the stem AND of model.1)."""
    import os
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", 1, seed=0)
    imgs = torch.from_numpy(synthetic_bscans(5, seed=34)).to(cuda_device)
    outs = []
    for off in (False, True):
        if off:
            os.environ["M355_NO_STEMFUSE"] = "1"
        try:
            eng = SegEngine("s", 1, (640, 640), max_batch=5)
        finally:
            os.environ.pop("M355_NO_STEMFUSE", None)
        kinds = [o["kernel"] for o in eng.op_infos()]
        assert any(k.startswith("stem+") for k in kinds) == (not off), kinds
        eng.load_state_dict(sd)
        p, q = eng.forward(imgs)
        raw = eng.raw_head(5)
        torch.cuda.synchronize()
        outs.append((p.clone(), q.clone(), raw.clone()))
        eng.close()
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())  # noqa: E731
    e_raw, e_pr = rel(outs[0][2], outs[1][2]), rel(outs[0][1], outs[1][1])
    print(f"stem fused vs two launches: raw head rel-L2 {e_raw:.2e}, prototypes rel-L2 {e_pr:.2e}")
    assert torch.isfinite(outs[0][2]).all()
    assert e_raw <= 2e-3 and e_pr <= 2e-3


def test_two_engines_in_flight_give_the_sequential_results(cuda_device):
    """bench.py's pipelining: two engine instances on two streams, forwards overlapping, each with its own activations, tile
    queues and weights.  Every output must equal, bit for bit, what one engine computes alone for the same batch -- whatever
    the overlap does to block placement and to the order in which persistent blocks claim their tiles."""
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", 1, seed=0)
    B = 8
    batches = [torch.from_numpy(synthetic_bscans(B, seed=70 + j)).to(cuda_device) for j in range(4)]
    ref_eng = SegEngine("s", 1, (640, 640), max_batch=B)
    ref_eng.load_state_dict(sd)
    refs = []
    for x in batches:
        p, q = ref_eng.forward(x)
        torch.cuda.synchronize()
        refs.append((p.clone(), q.clone()))
    ref_eng.close()
    engs = [SegEngine("s", 1, (640, 640), max_batch=B) for _ in range(2)]
    for e in engs:
        e.load_state_dict(sd)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None] * len(batches)
    for rep in range(3):                                   # several rounds: the queues are re-armed by every launch
        for j, x in enumerate(batches):
            e, s = engs[j % 2], streams[j % 2]
            with torch.cuda.stream(s):
                p, q = e.forward(x)
                outs[j] = (p.clone(), q.clone())            # (clone on the same stream: ordered after the forward)
        torch.cuda.synchronize()
        for j in range(len(batches)):
            assert torch.equal(outs[j][0], refs[j][0]) and torch.equal(outs[j][1], refs[j][1]), (rep, j)
    for e in engs:
        e.close()


def test_multi_label_postprocess_equals_the_oracle(cuda_device):
    """The validator's NMS mode (upstream `non_max_suppression(multi_label=True)`): every (anchor, class) pair
    above conf is a candidate.  Engine: one single-class NMS launch per class + merge; oracle: the joint class-offset NMS.
    Same preds on both sides -> identical rows; masks of the merged detections against the oracle's process_mask."""
    import yolov8_seg_oracle as orc
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    nc, B, imgsz = 3, 2, (320, 320)
    eng = SegEngine("n", nc, imgsz, max_batch=B)
    A = eng.num_anchors
    g = torch.Generator().manual_seed(11)
    cxy = torch.rand((B, A, 2), generator=g) * 280 + 20
    wh = torch.rand((B, A, 2), generator=g) * 80 + 8
    scores = torch.rand((B, A, nc), generator=g) ** 6                      # a few hundred pairs above 0.25, thousands above 0.001
    coefs = torch.randn((B, A, 32), generator=g)
    preds = torch.cat((cxy, wh, scores, coefs), -1).to(cuda_device)
    protos = torch.randn((B, imgsz[0] // 4, imgsz[1] // 4, 32), generator=g).half().to(cuda_device)
    for conf, iou, max_det in ((0.9, 0.7, 300), (0.25, 0.7, 300), (0.001, 0.6, 100)):     # below / at the max_det cut
        dets, counts, masks = eng.postprocess(preds, protos, conf, iou, max_det, masks=True, multi_label=True)
        torch.cuda.synchronize()
        ref = orc.non_max_suppression(preds.cpu().permute(0, 2, 1).numpy(), nc, conf, iou, max_det, multi_label=True)
        single = orc.non_max_suppression(preds.cpu().permute(0, 2, 1).numpy(), nc, conf, iou, max_det)
        for b in range(B):
            n = int(counts[b])
            assert n == ref[b].shape[0], (conf, n, ref[b].shape)
            assert np.array_equal(dets[b, :n].cpu().numpy(), ref[b]), (conf, b)
            d = dets[b, :n].cpu()
            want = orc.process_mask(protos[b].float().cpu().permute(2, 0, 1), d[:, 6:], d[:, :4], imgsz)
            assert float((masks[b, :n].cpu().bool() == want).float().mean()) >= 0.995
        print(f"conf {conf}: multi-label keeps {[int(c) for c in counts]}, best-class-per-anchor {[r.shape[0] for r in single]}")
        assert len({int(c) for c in dets[0, :int(counts[0]), 5].tolist()}) > 1          # several classes survive
    eng.close()
