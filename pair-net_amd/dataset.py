"""PSG ground truth for the evaluator and the loss forward: the annotation side of the
reference's dataset class, with the per-pixel work on the GPU.

What the reference does on the host for every test image before `sgg_evaluation`
(pairnet/datasets/psg.py:345-388) -- read the panoptic PNG, `rgb2id`, one `seg_map == id`
boolean map per annotated segment -- and its training-side loader
(`LoadPanopticSceneGraphAnnotations._load_masks_and_semantic_segs`,
pairnet/datasets/pipelines/loading.py:110-158) is ONE kernel here (`pn_pan_masks_u8`: 3 bytes
read, G bytes written per pixel); the masks stay on the device, where `TripletEvaluator` and
`CrossHead2.loss` consume them.  The list logic of `PanopticSceneGraphDataset.__init__`
(:62-92) and `get_ann_info` (:180-272) is host code on the loaded json, as in the reference.

Decoding the PNG / reading the json file is the caller's I/O (PIL, cv2, mmcv ...: any decoder
that yields the (H, W, 3) uint8 RGB array).
"""
import json
import random
from collections import defaultdict

import numpy as np
import torch

from . import hip


def load_psg(ann_file, split="test"):
    """`PanopticSceneGraphDataset.__init__`'s view of the annotation file (psg.py:62-110):
    predicate labels become 1-based, images without relations are dropped, the split is taken
    by `test_image_ids`.  `ann_file`: a path or the loaded dict -- left untouched: the entries
    returned are shallow copies with new relation lists (the reference re-reads the file per
    dataset object, so it never shifts the labels twice).  Returns dict(data, classes,
    predicates)."""
    if isinstance(ann_file, (str, bytes)) or hasattr(ann_file, "__fspath__"):
        with open(ann_file) as f:
            dataset = json.load(f)
    else:
        dataset = ann_file
    if split not in ("train", "test"):
        raise ValueError("split: 'train' or 'test'")
    data = [dict(d, relations=[[r[0], r[1], r[2] + 1] for r in d["relations"]])
            for d in dataset["data"] if len(d["relations"]) != 0]
    test_ids = dataset["test_image_ids"]
    keep = (lambda d: d["image_id"] in test_ids) if split == "test" else \
        (lambda d: d["image_id"] not in test_ids)
    return dict(data=[d for d in data if keep(d)],
                classes=list(dataset["thing_classes"]) + list(dataset["stuff_classes"]),
                predicates=list(dataset["predicate_classes"]))


def ann_info(d, split="test", all_bboxes=False):
    """`get_ann_info` (psg.py:180-272) of one entry of `load_psg(...)["data"]`: boxes and labels
    (things only unless `all_bboxes`), the segment list, relations without exact duplicates
    (test) or with one random predicate per pair (train, `np.random.choice` like the
    reference), the relation map (`random.random()` where a pair carries several predicates)."""
    if all_bboxes:
        gt_bboxes = np.array([a["bbox"] for a in d["annotations"]], dtype=np.float32)
        gt_labels = np.array([a["category_id"] for a in d["annotations"]], dtype=np.int64)
    else:
        pairs = [(a["bbox"], a["category_id"]) for a, s in zip(d["annotations"], d["segments_info"])
                 if s["isthing"]]
        gt_bboxes = np.array([b for b, _ in pairs], dtype=np.float32) if pairs else \
            np.zeros((0, 4), dtype=np.float32)
        gt_labels = np.array([c for _, c in pairs], dtype=np.int64)
    masks = [dict(id=s["id"], category=s["category_id"], is_thing=s["isthing"])
             for s in d["segments_info"]]
    if split == "train":
        by_pair = defaultdict(list)
        for o0, o1, r in d["relations"]:
            by_pair[(o0, o1)].append(r)
        rels = np.array([(k[0], k[1], np.random.choice(v)) for k, v in by_pair.items()],
                        dtype=np.int32)
    else:
        uniq = []
        for o0, o1, r in d["relations"]:
            if (o0, o1, r) not in uniq:
                uniq.append((o0, o1, r))
        rels = np.array(uniq, dtype=np.int32)
    rel_map = np.zeros((len(masks), len(masks)), dtype=np.int64)
    for s, o, r in rels.tolist():
        if rel_map[s, o] > 0:
            if random.random() > 0.5:
                rel_map[s, o] = r
        else:
            rel_map[s, o] = r
    return dict(bboxes=gt_bboxes, labels=gt_labels, rels=rels, rel_maps=rel_map,
                bboxes_ignore=np.zeros((0, 4), dtype=np.float32), masks=masks,
                seg_map=d["pan_seg_file_name"])


def _segments(ann, device):
    ids = torch.tensor([m["id"] for m in ann["masks"]], dtype=torch.int32, device=device)
    cats = torch.tensor([m["category"] for m in ann["masks"]], dtype=torch.int32, device=device)
    return ids, cats


def _rgb(pan_rgb, device):
    rgb = torch.as_tensor(np.ascontiguousarray(pan_rgb) if isinstance(pan_rgb, np.ndarray)
                          else pan_rgb)
    if rgb.dtype != torch.uint8 or rgb.dim() != 3 or rgb.shape[2] != 3:
        raise ValueError("the panoptic PNG as an (H, W, 3) uint8 RGB array, got %s %s"
                         % (tuple(rgb.shape), rgb.dtype))
    return rgb.to(device).contiguous()


@torch.no_grad()
def eval_ground_truth(ann, pan_rgb, device):
    """One image's ground truth as the evaluation loop needs it (psg.py:345-388): labels made
    1-based (`ann["labels"] += 1`: a copy here, `ann` is left alone), relations, and one
    boolean mask per annotated segment [G, H, W] ON THE DEVICE.  The keys are the ones
    `dist.multi_gpu_test(annotations=...)` and `TripletEvaluator` take."""
    device = torch.device(device)
    rgb = _rgb(pan_rgb, device)
    ids, cats = _segments(ann, device)
    H, W = int(rgb.shape[0]), int(rgb.shape[1])
    masks = torch.empty((ids.shape[0], H, W), dtype=torch.bool, device=device)
    with torch.cuda.device(device):
        hip.pan_masks(rgb, ids, None, masks)
    return dict(gt_rels=ann["rels"], gt_labels=ann["labels"] + 1, gt_masks=masks,
                gt_bboxes=ann["bboxes"], rel_maps=ann["rel_maps"])


@torch.no_grad()
def load_masks_and_semantic_seg(ann, pan_rgb, device):
    """The training-side loader (loading.py:110-158): (gt_masks [G, H, W] uint8 -- every
    segment, things and stuff --, gt_semantic_seg [H, W] int32 with 255 = unlabelled), both on
    the device (what `PSGTr.val_losses` takes as `gt_masks`)."""
    device = torch.device(device)
    rgb = _rgb(pan_rgb, device)
    ids, cats = _segments(ann, device)
    H, W = int(rgb.shape[0]), int(rgb.shape[1])
    masks = torch.empty((ids.shape[0], H, W), dtype=torch.uint8, device=device)
    sem = torch.empty((H, W), dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        hip.pan_masks(rgb, ids, cats, masks, sem)
    return masks, sem
