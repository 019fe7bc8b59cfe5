"""GPU: the ground-truth kernel (`pn_pan_masks_u8`, pair-net_amd/dataset.py) against
oracle/dataset.py (pinned bit for bit to the reference's dataset class and loader on the CPU,
tests/test_dataset.py): byte work, exact equality; and the ground truth it produces through the
device evaluator."""
import copy

import numpy as np
import pytest
import torch

from oracle import dataset as D
from test_dataset import synthetic_psg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("H,W", [(37, 53), (40, 64), (1, 1), (3, 5), (480, 640)])
def test_ground_truth_masks_equal_the_oracle_bit_for_bit(H, W):
    from pairnet_amd import dataset as P
    dataset, images = synthetic_psg(H * 1000 + W, n_images=5, H=H, W=W)
    data = P.load_psg(copy.deepcopy(dataset), "test")["data"]
    assert data
    for d in data:
        ann = P.ann_info(d)
        rgb = images[d["pan_seg_file_name"]]
        want = D.eval_groundtruth(copy.deepcopy(ann), rgb)
        got = P.eval_ground_truth(ann, rgb, DEV)
        assert np.array_equal(got["gt_labels"], want["labels"]) and np.array_equal(got["gt_rels"], want["rels"])
        assert np.array_equal(ann["labels"] + 1, want["labels"])           # (ann itself untouched)
        assert got["gt_masks"].dtype == torch.bool and got["gt_masks"].shape == (len(want["masks"]), H, W)
        assert np.array_equal(got["gt_masks"].cpu().numpy(), np.stack(want["masks"], 0))
        m, sem = P.load_masks_and_semantic_seg(ann, torch.from_numpy(rgb), DEV)
        wm, wsem = D.load_masks_and_semantic_seg(ann, rgb)
        assert np.array_equal(m.cpu().numpy(), wm) and m.dtype == torch.uint8
        assert np.array_equal(sem.cpu().numpy(), wsem)


def test_ground_truth_edge_cases():
    from pairnet_amd import dataset as P
    from pairnet_amd import hip
    rng = np.random.RandomState(0)
    rgb = rng.randint(0, 3, (11, 7, 3)).astype(np.uint8)
    # no segments: an empty mask stack, the semantic map all 255
    ann = dict(masks=[], rels=np.zeros((0, 3), np.int32), labels=np.zeros(0, np.int64),
               bboxes=np.zeros((0, 4), np.float32), rel_maps=np.zeros((0, 0), np.int64))
    m, sem = P.load_masks_and_semantic_seg(ann, rgb, DEV)
    assert m.shape == (0, 11, 7) and bool((sem == 255).all())
    # the same id listed twice: both masks equal, the LATER category wins (np.where in order)
    sid = int(D.rgb2id(rgb)[0, 0])
    ann["masks"] = [dict(id=sid, category=3, is_thing=1), dict(id=sid, category=7, is_thing=0),
                    dict(id=2 ** 24 - 1, category=1, is_thing=1)]
    m, sem = P.load_masks_and_semantic_seg(ann, rgb, DEV)
    wm, wsem = D.load_masks_and_semantic_seg(ann, rgb)
    assert np.array_equal(m.cpu().numpy(), wm) and np.array_equal(sem.cpu().numpy(), wsem)
    assert int(m[2].sum()) == 0 and int(sem[0, 0]) == 7
    with pytest.raises(ValueError):
        P.eval_ground_truth(ann, rgb.astype(np.float32), DEV)
    too_many = torch.zeros(257, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        hip.pan_masks(torch.from_numpy(rgb).to(DEV), too_many, None,
                      torch.empty((257, 11, 7), dtype=torch.uint8, device=DEV))


def test_ground_truth_feeds_the_device_evaluator():
    """A prediction that IS the ground truth (its masks, labels and relations) scores recall 1
    through `TripletEvaluator` when the ground truth comes from `eval_ground_truth`."""
    from pairnet_amd import TripletEvaluator
    from pairnet_amd import dataset as P
    dataset, images = synthetic_psg(21, n_images=4, H=48, W=64)
    d = P.load_psg(copy.deepcopy(dataset), "test")["data"][0]
    ann = P.ann_info(d, all_bboxes=True)
    gt = P.eval_ground_truth(ann, images[d["pan_seg_file_name"]], DEV)
    rels = gt["gt_rels"]
    R, C1 = len(rels), 8
    masks = torch.cat([gt["gt_masks"][rels[:, 0].tolist()], gt["gt_masks"][rels[:, 1].tolist()]], 0)
    labels = torch.from_numpy(np.concatenate([gt["gt_labels"][rels[:, 0]], gt["gt_labels"][rels[:, 1]]])).to(DEV)
    r_dists = torch.zeros(R, C1, device=DEV)
    r_dists[torch.arange(R), torch.from_numpy(rels[:, 2].astype(np.int64))] = 1.0
    result = (None, labels, None, masks, None, None, None, r_dists)
    ev = TripletEvaluator()(result, gt["gt_rels"], gt["gt_labels"], gt["gt_masks"])
    hit = {g for row in ev["pred_to_gt"] for g in row}
    assert hit == set(range(R))
