"""CPU restatement of the reference's PSG ground-truth preparation: what the evaluation loop and
the loss forward are fed with besides the detector's results (the data
formats on the ground-truth side of the path).

TEST INFRASTRUCTURE: only tests/ may import this.  Follows, and is pinned bit for bit
(tests/test_dataset.py: the reference's own methods executed from /root/reference under
oracle/ref_shim.install_dataset(), same seeded synthetic annotation file, same `random` state):

  PanopticSceneGraphDataset.__init__    pairnet/datasets/psg.py:62-110   (`load_psg`: predicate
                                        labels made 1-based, images without relations dropped,
                                        the train / test split)
  PanopticSceneGraphDataset.get_ann_info  psg.py:180-272                  (`ann_info`)
  PanopticSceneGraphDataset.evaluate    psg.py:345-388                    (`eval_groundtruth`:
                                        labels made 1-based, one boolean mask per segment)
  LoadPanopticSceneGraphAnnotations._load_masks_and_semantic_segs
                                        pairnet/datasets/pipelines/loading.py:110-158
                                        (`load_masks_and_semantic_seg`)

`rgb2id` is panopticapi's ([3P], absent from /root/reference and from the image; its published
definition: id = R + 256 G + 256^2 B on the RGB panoptic PNG).
"""
import random
from collections import defaultdict

import numpy as np


def rgb2id(color):
    """panopticapi.utils.rgb2id for an (H, W, 3) uint8 array (-> int32) or one colour."""
    if isinstance(color, np.ndarray) and color.ndim == 3:
        c = color.astype(np.int32) if color.dtype == np.uint8 else color
        return c[:, :, 0] + 256 * c[:, :, 1] + 256 * 256 * c[:, :, 2]
    return int(color[0] + 256 * color[1] + 256 * 256 * color[2])


def load_psg(dataset, split):
    """psg.py:62-92 on the loaded annotation dict (mutates the relations like the reference:
    predicate labels become 1-based)."""
    for d in dataset["data"]:
        for r in d["relations"]:
            r[2] += 1
    data = [d for d in dataset["data"] if len(d["relations"]) != 0]
    assert split in {"train", "test"}
    test_ids = dataset["test_image_ids"]
    if split == "train":
        return [d for d in data if d["image_id"] not in test_ids]
    return [d for d in data if d["image_id"] in test_ids]


def ann_info(d, split="test", all_bboxes=False):
    """psg.py:180-272.  Uses `random` / `np.random` exactly where the reference does (duplicate
    subject-object pairs), in the same order."""
    gt_bboxes_ignore = np.zeros((0, 4), dtype=np.float32)
    if all_bboxes:
        gt_bboxes = np.array([a["bbox"] for a in d["annotations"]], dtype=np.float32)
        gt_labels = np.array([a["category_id"] for a in d["annotations"]], dtype=np.int64)
    else:
        boxes, labels = [], []
        for a, s in zip(d["annotations"], d["segments_info"]):
            if s["isthing"]:
                boxes.append(a["bbox"])
                labels.append(a["category_id"])
        if boxes:
            gt_bboxes = np.array(boxes, dtype=np.float32)
            gt_labels = np.array(labels, dtype=np.int64)
        else:
            gt_bboxes = np.zeros((0, 4), dtype=np.float32)
            gt_labels = np.array([], dtype=np.int64)
    masks = [{"id": s["id"], "category": s["category_id"], "is_thing": s["isthing"]}
             for s in d["segments_info"]]
    gt_rels = d["relations"].copy()
    if split == "train":        # one random predicate per (subject, object) pair
        sets = defaultdict(list)
        for o0, o1, r in gt_rels:
            sets[(o0, o1)].append(r)
        gt_rels = np.array([(k[0], k[1], np.random.choice(v)) for k, v in sets.items()],
                           dtype=np.int32)
    else:                       # exact duplicates dropped, several predicates per pair kept
        seen = []
        for o0, o1, r in gt_rels:
            if (o0, o1, r) not in seen:
                seen.append((o0, o1, r))
        gt_rels = np.array(seen, dtype=np.int32)
    n = len(masks)
    rel_map = np.zeros((n, n), dtype=np.int64)
    for i in range(gt_rels.shape[0]):
        s, o, r = int(gt_rels[i, 0]), int(gt_rels[i, 1]), int(gt_rels[i, 2])
        if rel_map[s, o] > 0:
            if random.random() > 0.5:
                rel_map[s, o] = r
        else:
            rel_map[s, o] = r
    return dict(bboxes=gt_bboxes, labels=gt_labels, rels=gt_rels, rel_maps=rel_map,
                bboxes_ignore=gt_bboxes_ignore, masks=masks, seg_map=d["pan_seg_file_name"])


def eval_groundtruth(ann, pan_rgb):
    """psg.py:345-388: one image's ground truth as `sgg_evaluation` receives it (the fields of
    the `Result` the reference builds).  `ann` = ann_info(d); `pan_rgb` the decoded (H, W, 3)
    RGB panoptic PNG.  Mutates ann["labels"] like the reference (1-based)."""
    ann["labels"] += 1
    seg = rgb2id(pan_rgb.copy())
    masks = [seg == s["id"] for s in ann["masks"]]
    return dict(bboxes=ann["bboxes"], labels=ann["labels"], rels=ann["rels"],
                relmaps=ann["rel_maps"], rel_pair_idxes=ann["rels"][:, :2],
                rel_labels=ann["rels"][:, -1], masks=masks)


def load_masks_and_semantic_seg(ann, pan_rgb):
    """loading.py:110-158: (gt_masks [G, H, W] uint8 -- every segment, things and stuff --,
    gt_semantic_seg [H, W] with 255 = unlabelled) from the RGB panoptic PNG."""
    seg = rgb2id(pan_rgb)
    gt_seg = np.zeros_like(seg) + 255
    masks = []
    for m in ann["masks"]:
        mask = seg == m["id"]
        gt_seg = np.where(mask, m["category"], gt_seg)
        masks.append(mask.astype(np.uint8))
    h, w = seg.shape
    gt_masks = np.stack(masks, 0) if masks else np.zeros((0, h, w), dtype=np.uint8)
    return gt_masks, gt_seg
