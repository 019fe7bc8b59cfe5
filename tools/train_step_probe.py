"""What one training iteration of the head behind the pixel decoder costs on one MI355X
(pair-net_amd/train.py): ms per `TailTrainer.step` at 800x1333, one image, for the
tail alone, with the nine masked decoder layers, and with the pixel decoder's encoder path as
well (everything behind the frozen backbone), and from the image with the ResNet-50's stages 2-4
("all": the reference's whole trainable graph), and where the time goes (HIP events around the
phases of a step; the Hungarian assignments inside `loss` are host work as in the reference).
Prints one JSON line.  `rocprofv3 --kernel-trace --stats -- python tools/train_step_probe.py`
gives the per-kernel view (profiles/r06_train_step_kernel_stats.csv)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pairnet_amd import CrossHead2, ResNet50Hip, TailTrainer, pairnet_head_cfg  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = pairnet_head_cfg()
cfg.pop("type")
head = CrossHead2(**cfg)
head.init_weights(seed=0)
head.to(dev)
net = ResNet50Hip().to(dev)
B, H, W = 1, 800, 1333
g = torch.Generator().manual_seed(3)
img = torch.randn(B, 3, H, W, generator=g).to(dev)
feats = [f.clone(memory_format=torch.preserve_format) for f in net(img)]      # frozen backbone
metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)] * B
G, T = 12, 10
Hh, Wh = 4 * feats[0].shape[2] // 2, 4 * feats[0].shape[3] // 2
gt_masks = [(torch.rand(G, Hh // 8, Wh // 8, generator=g) > 0.7).to(dev)
            .repeat_interleave(8, 1).repeat_interleave(8, 2).contiguous() for _ in range(B)]
gt_labels = [torch.randint(0, head.num_classes, (G,), generator=g) for _ in range(B)]
gt_rels = [torch.stack([torch.randint(0, G, (T,), generator=g), torch.randint(0, G, (T,), generator=g),
                        torch.randint(1, head.num_relations + 1, (T,), generator=g)], 1)
           for _ in range(B)]
pts = [torch.rand(1, 12544, 2, generator=g) for _ in range(B)]

out = {"what": "TailTrainer.step, 800x1333, one image, frozen ResNet-50 features resident in HBM "
               "(scope `all`: from the image, backbone stages 2-4 trained; its taped forward is "
               "counted under `backward`); "
               "ms per step over %d steps (device wait at both ends) and the phases of one step "
               "(HIP events; `loss` includes the two Hungarian assignments on the host)" % steps}
for scope in ("tail", "head", "head+pixel_decoder", "all"):
    mode = scope != "tail"
    tr = TailTrainer(head, train_decoder=mode, train_pixel_decoder=scope == "head+pixel_decoder",
                     backbone=net if scope == "all" else None)
    inp = img if scope == "all" else feats
    for _ in range(3):
        vals = tr.step(inp, metas, gt_rels, gt_labels, gt_masks, point_coords=pts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        vals = tr.step(inp, metas, gt_rels, gt_labels, gt_masks, point_coords=pts)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    # phases of one step
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    nh = tr.tape.flat_numel
    with torch.no_grad():
        ev[0].record()
        if scope == "all":
            feats = [f.clone(memory_format=torch.preserve_format) for f in net(img)]
        outs = head.forward(feats, metas)
        ev[1].record()
        up = {}
        head.loss(*outs, gt_rels, None, gt_labels, gt_masks, metas, point_coords=pts, grads=up)
        ev[2].record()
        pl = head._last_plan
        if mode:
            tr.tape.forward_from_plan(pl, pl.sub_pos, pl.obj_pos)
        else:
            tr.tape.forward(pl.q, pl.sub_pos, pl.obj_pos)
        if tr.pd_tape is not None:
            tr.pd_tape.forward(feats)
        ev[3].record()
        back = tr.tape.backward(g_rel=up["rel"], g_importance=up["importance"])
        if tr.pd_tape is not None:
            dfeats, _ = tr.pd_tape.backward(back[0])
            if tr.bb_tape is not None:
                tr.bb_tape.forward(feats[0])
                tr.bb_tape.backward(dfeats[2], dfeats[1], dfeats[0])
        ev[4].record()
        tr.apply_gradients()
        ev[5].record()
    torch.cuda.synchronize()
    ph = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    out[scope] = {
        "ms_per_step": ms, "trained_parameters": int(sum(v.numel() for v in tr.params.values())),
        "phases_ms": dict(zip(("inference_forward", "loss_and_logit_gradients", "taped_forward",
                               "backward", "clip_adamw_refresh"), ph)),
        "loss": {k: float(v) for k, v in vals.items()}}
print(json.dumps(out))
