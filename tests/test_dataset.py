"""CPU: oracle/dataset.py against the reference's own dataset code executed from
/root/reference (oracle/ref_shim.install_dataset: psg.py and pipelines/loading.py run in place
under name-only stubs) on a seeded synthetic PSG annotation file -- bit for bit, integer / byte
work.  Skipped where the reference tree is absent (the GPU box)."""
import copy
import json
import random

import numpy as np
import pytest

from oracle import dataset as D
from oracle import ref_shim


def synthetic_psg(seed, n_images=7, H=37, W=53):
    """A PSG-format annotation dict + its panoptic PNGs (as RGB arrays): things and stuff, ids
    above 2^16 (all three colour bytes matter), a segment listed in the annotations but absent
    from the PNG, duplicate relations, several predicates per pair, an image without
    relations."""
    rng = np.random.RandomState(seed)
    thing, stuff = ["person", "dog", "car"], ["sky", "grass"]
    data, images = [], {}
    for i in range(n_images):
        G = int(rng.randint(2, 7))
        ids = [int(x) for x in rng.choice(np.arange(1, 2 ** 24 - 1), G, replace=False)]
        grid = rng.randint(0, G if i % 3 else G - 1, (H, W))        # (i % 3 == 0: last id unused)
        seg = np.asarray(ids, dtype=np.int64)[grid]
        rgb = np.stack([seg % 256, (seg // 256) % 256, seg // 65536], -1).astype(np.uint8)
        name = "pan_%03d.png" % i
        images[name] = rgb
        cats = [int(c) for c in rng.randint(0, 5, G)]
        segs = [dict(id=ids[g], category_id=cats[g], iscrowd=0, isthing=int(cats[g] < 3),
                     area=int((seg == ids[g]).sum())) for g in range(G)]
        anns = [dict(bbox=[float(v) for v in np.sort(rng.rand(4) * 30)], category_id=cats[g])
                for g in range(G)]
        rels = []
        if i != 4:                                                     # image 4: no relations
            for _ in range(int(rng.randint(1, 9))):
                s, o = (int(x) for x in rng.choice(G, 2, replace=False))
                rels.append([s, o, int(rng.randint(0, 6))])
            rels.append(list(rels[0]))                                 # an exact duplicate
            rels.append([rels[0][0], rels[0][1], (rels[0][2] + 1) % 6])  # same pair, other label
        data.append(dict(file_name="img_%03d.jpg" % i, height=H, width=W, image_id=str(100 + i),
                         pan_seg_file_name=name, segments_info=segs, annotations=anns,
                         relations=rels))
    dataset = dict(data=data, thing_classes=thing, stuff_classes=stuff,
                   predicate_classes=["p%d" % k for k in range(6)],
                   test_image_ids=[d["image_id"] for d in data[1::2]] + [d["image_id"] for d in data[4:5]])
    return dataset, images


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
@pytest.mark.parametrize("split", ["test", "train"])
@pytest.mark.parametrize("all_bboxes", [False, True])
def test_oracle_ground_truth_equals_the_reference_dataset(tmp_path, split, all_bboxes):
    Dataset, Loader = ref_shim.install_dataset()
    dataset, images = synthetic_psg(3)
    ref_shim.PAN_IMAGES.clear()
    ref_shim.PAN_IMAGES.update(images)
    path = tmp_path / "psg.json"
    path.write_text(json.dumps(dataset))
    ref = Dataset(str(path), pipeline=[], seg_prefix="seg", split=split, all_bboxes=all_bboxes,
                  test_mode=True)
    mine = D.load_psg(copy.deepcopy(dataset), split)
    assert [d["image_id"] for d in mine] == [d["image_id"] for d in ref.data] and len(mine) >= 2
    assert json.dumps(mine, sort_keys=True) == json.dumps(ref.data, sort_keys=True)
    loader = Loader(with_rel=True)
    for i, d in enumerate(mine):
        random.seed(11 + i); np.random.seed(11 + i)
        want = ref.get_ann_info(i)
        random.seed(11 + i); np.random.seed(11 + i)
        got = D.ann_info(d, split=split, all_bboxes=all_bboxes)
        assert set(got) == set(want)
        for k in want:
            if isinstance(want[k], np.ndarray):
                assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k
            else:
                assert got[k] == want[k], k
        # the training-side loader of the same annotation (loading.py:110-158)
        res = dict(ann_info=want, seg_prefix="seg", img_info=dict(height=d["height"], width=d["width"]),
                   mask_fields=[], seg_fields=[])
        loader._load_masks_and_semantic_segs(res)
        masks, sem = D.load_masks_and_semantic_seg(got, images[d["pan_seg_file_name"]])
        assert masks.dtype == np.uint8 and np.array_equal(masks, res["gt_masks"].to_ndarray())
        assert sem.dtype == res["gt_semantic_seg"].dtype and np.array_equal(sem, res["gt_semantic_seg"])
    if split != "test":
        return
    # the ground truth `evaluate` builds for sgg_evaluation (psg.py:345-388)
    random.seed(5); np.random.seed(5)
    assert ref.evaluate([], metric="sgdet") == "captured"
    gts = ref_shim.CAPTURED["groundtruths"]
    random.seed(5); np.random.seed(5)
    assert len(gts) == len(mine)
    for d, gt in zip(mine, gts):
        got = D.eval_groundtruth(D.ann_info(d, split="test", all_bboxes=all_bboxes),
                                 images[d["pan_seg_file_name"]])
        assert set(got) == set(gt.__dict__)
        for k, v in got.items():
            w = getattr(gt, k)
            if k == "masks":
                assert len(v) == len(w) and all(a.dtype == np.bool_ and np.array_equal(a, b)
                                                for a, b in zip(v, w))
            else:
                assert v.dtype == w.dtype and np.array_equal(v, w), k
    assert ref_shim.CAPTURED["ind_to_predicates"] == ["__background__"] + dataset["predicate_classes"]


def test_rgb2id_is_little_endian_over_the_colour_bytes():
    rgb = np.array([[[1, 2, 3], [255, 255, 255], [0, 0, 1]]], dtype=np.uint8)
    assert D.rgb2id(rgb).tolist() == [[1 + 2 * 256 + 3 * 65536, 2 ** 24 - 1, 65536]]
    assert D.rgb2id(rgb).dtype == np.int32 and D.rgb2id((7, 1, 0)) == 263


@pytest.mark.parametrize("split", ["test", "train"])
@pytest.mark.parametrize("all_bboxes", [False, True])
def test_product_host_logic_equals_the_oracle(split, all_bboxes):
    """pairnet_amd.dataset.load_psg / ann_info (host code) == oracle/dataset.py, which the test
    above pins to the reference's dataset class."""
    from pairnet_amd import dataset as P
    dataset, _ = synthetic_psg(9, n_images=9)
    want = D.load_psg(copy.deepcopy(dataset), split)
    got = P.load_psg(copy.deepcopy(dataset), split)
    assert json.dumps(got["data"], sort_keys=True) == json.dumps(want, sort_keys=True)
    assert got["classes"] == dataset["thing_classes"] + dataset["stuff_classes"]
    assert got["predicates"] == dataset["predicate_classes"]
    for i, d in enumerate(want):
        random.seed(i); np.random.seed(i)
        a = D.ann_info(d, split=split, all_bboxes=all_bboxes)
        random.seed(i); np.random.seed(i)
        b = P.ann_info(got["data"][i], split=split, all_bboxes=all_bboxes)
        assert set(a) == set(b)
        for k in a:
            if isinstance(a[k], np.ndarray):
                assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
            else:
                assert a[k] == b[k], k
    with pytest.raises(ValueError):
        P.load_psg(copy.deepcopy(dataset), "val")


def test_load_psg_leaves_the_callers_dict_alone():
    """loading both splits from ONE loaded dict must not shift the predicate labels
    twice (the reference re-reads the file per dataset object)."""
    from pairnet_amd import dataset as P
    dataset, _ = synthetic_psg(9, n_images=9)
    d = copy.deepcopy(dataset)
    test = P.load_psg(d, "test")
    assert d == dataset
    again = P.load_psg(d, "test")
    assert again["data"] == test["data"]
    train = P.load_psg(d, "train")
    want = D.load_psg(copy.deepcopy(dataset), "train")
    assert [x["relations"] for x in train["data"]] == [x["relations"] for x in want]


def test_load_psg_reads_a_file(tmp_path):
    from pairnet_amd import dataset as P
    dataset, _ = synthetic_psg(2)
    f = tmp_path / "psg.json"
    f.write_text(json.dumps(dataset))
    assert len(P.load_psg(str(f), "test")["data"]) == len(D.load_psg(copy.deepcopy(dataset), "test"))
