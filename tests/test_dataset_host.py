"""Dataset / label pipeline host logic (SURVEY.md A2, N2 converter): yaml parsing, polygon labels, letterboxed
geometry, overlap masks, epoch sharding across ranks, annotations.json -> polygon dataset.  CPU only."""
import json
import os

import numpy as np
import pytest

from defectdetection_viaobjectdetection_amd import dataset as D

HERE = os.path.dirname(os.path.abspath(__file__))


def _write_image(path, h, w, value=90):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(np.full((h, w), value, np.uint8)).save(path)


def test_yaml_and_label_paths(tmp_path):
    root = tmp_path / "ds"
    _write_image(str(root / "images" / "train" / "a.png"), 40, 80)
    os.makedirs(root / "labels" / "train")
    (root / "labels" / "train" / "a.txt").write_text("0 0.25 0.25 0.75 0.25 0.75 0.75 0.25 0.75\n0 0.5 0.5 0.2 0.2\n")
    y = tmp_path / "d.yaml"
    y.write_text("path: ds\ntrain: images/train\nnames:\n  0: defect\n")
    cfg = D.read_data_yaml(str(y))
    assert cfg["nc"] == 1 and cfg["names"] == {0: "defect"} and cfg["val"] == cfg["train"]
    assert D.img2label_path(str(root / "images" / "train" / "a.png")) == str(root / "labels" / "train" / "a.txt")
    ds = D.SegDataset(cfg["train"], 64, nc=1)
    assert len(ds) == 1 and ds.images.shape == (1, 64, 64, 3)
    # 40x80 -> ratio 0.8 -> 32x64, padded 16 rows top/bottom: polygon x 0.25*64 = 16, y 0.25*32 + 16 = 24
    c, poly = ds.labels[0][0]
    np.testing.assert_allclose(poly[0], [16, 24], atol=1e-9)
    np.testing.assert_allclose(poly[2], [48, 40], atol=1e-9)
    b = ds.batch([0])
    assert b["img"].dtype == np.uint8 and b["masks"].shape == (1, 16, 16)
    assert b["bboxes"].shape == (2, 4) and b["batch_idx"].tolist() == [0, 0]
    # instances sorted by area descending: the big polygon is value 1, the small box (centre) overwrites with 2
    assert b["masks"][0, 8, 8] == 2 and b["masks"][0, 6, 5] == 1 and b["masks"][0, 0, 0] == 0
    np.testing.assert_allclose(b["bboxes"][0], [0.5, 0.5, 0.5, 0.25], atol=1e-6)
    f = ds.batch([0], flip=[True])
    np.testing.assert_allclose(f["bboxes"][0], [0.5, 0.5, 0.5, 0.25], atol=1e-6)      # symmetric case stays put
    with pytest.raises(ValueError):
        (root / "labels" / "train" / "a.txt").write_text("0 0.1 0.2 0.3\n")
        D.SegDataset(cfg["train"], 64)


def test_flip_moves_boxes(tmp_path):
    root = tmp_path / "ds"
    _write_image(str(root / "images" / "train" / "a.png"), 64, 64)
    os.makedirs(root / "labels" / "train")
    (root / "labels" / "train" / "a.txt").write_text("0 0.0 0.0 0.25 0.0 0.25 0.5 0.0 0.5\n")
    ds = D.SegDataset(str(root / "images" / "train"), 64)
    b, f = ds.batch([0]), ds.batch([0], flip=[True])
    np.testing.assert_allclose(b["bboxes"][0], [0.125, 0.25, 0.25, 0.5], atol=1e-6)
    np.testing.assert_allclose(f["bboxes"][0], [0.875, 0.25, 0.25, 0.5], atol=1e-6)
    assert (f["masks"][0] == b["masks"][0][:, ::-1]).all()


def test_epoch_batches_shard_without_overlap():
    n, batch, world = 37, 4, 2
    per_rank = [D.epoch_batches(n, batch, 3, seed=0, rank=r, world=world) for r in range(world)]
    assert len(per_rank[0]) == len(per_rank[1]) == 5
    for s in range(5):
        assert len(per_rank[0][s]) == len(per_rank[1][s]) == batch
        assert not set(per_rank[0][s]) & set(per_rank[1][s]) or s == 4      # only the wrapped tail may repeat
    seen = {i for r in per_rank for b in r for i in b}
    assert seen == set(range(n))
    assert D.epoch_batches(n, batch, 3, 0, 0, 2) == per_rank[0] and D.epoch_batches(n, batch, 4, 0, 0, 2) != per_rank[0]


def test_annotations_converter_handles_reversed_x(tmp_path):
    ann = json.load(open(os.path.join(HERE, "golden", "annotations_excerpt.json")))["annotations"]
    for fo, files in ann.items():
        for fi in files:
            _write_image(str(tmp_path / "raw" / fo / fi), 200, 320)
    ypath = D.write_polygon_dataset(ann, str(tmp_path / "raw"), str(tmp_path / "out"), val_fraction=0.34, seed=1)
    cfg = D.read_data_yaml(ypath)
    assert cfg["names"] == {0: "Delamination"}
    tr, va = D.SegDataset(cfg["train"], 320), D.SegDataset(cfg["val"], 320)
    assert len(tr) + len(va) == sum(len(f) for f in ann.values()) and len(va) == 2
    for ds in (tr, va):
        for inst in ds.labels:
            assert inst, "every excerpt image has at least one usable box"
            for _, p in inst:
                assert p[:, 0].max() > p[:, 0].min() and p[:, 1].max() > p[:, 1].min()   # x_min > x_max input is repaired


def test_overlap_mask_switches_to_int32_beyond_255_instances():
    """A.4: the overlap map stores instance index + 1; with more than 255 instances uint8 would wrap around (upstream
    switches to int32 there as well)."""
    from defectdetection_viaobjectdetection_amd.dataset import overlap_mask
    polys = []
    for i in range(300):                                    # 300 squares of 8 x 8 px on a 20 x 15 grid, 160 x 256 map at ratio 4... of a 640 x 1024 image
        y, x = (i // 20) * 40, (i % 20) * 48
        polys.append(np.array([[x, y], [x + 32, y], [x + 32, y + 32], [x, y + 32]], np.float32))
    m, order = overlap_mask(polys, (640, 1024))
    assert m.dtype == np.int32 and len(order) == 300
    assert sorted(np.unique(m).tolist()) == list(range(0, 301))
    m2, _ = overlap_mask(polys[:200], (640, 1024))
    assert m2.dtype == np.uint8 and m2.max() == 200
