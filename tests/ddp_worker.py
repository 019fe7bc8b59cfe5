"""Worker of tests/test_train_gpu.py::test_data_parallel_training_*: one rank of a data-parallel
training run on cuda:0 (two gloo ranks share the test box's single GPU; `nccl` with one rank runs
the real RCCL all-reduce).  usage: ddp_worker.py RANK WORLD PORT BACKEND OUT [noapply]"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def batch(seed, H=96, W=128):
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn(1, c, H // s, W // s, generator=g).cuda()
             for c, s in zip((256, 512, 1024, 2048), (4, 8, 16, 32))]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4)]
    gt_labels = [torch.tensor([3, 17, 90, 120, 3])]
    gt_masks = [torch.rand(5, H, W, generator=g) > 0.6]
    gt_rels = [torch.tensor([[0, 1, 5], [2, 3, 17], [1, 0, 56], [4, 2, 5], [0, 1, 9]])]
    pts = [torch.rand(1, 12544, 2, generator=g)]
    return feats, metas, gt_rels, gt_labels, gt_masks, pts


def main():
    rank, world, port, backend, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    noapply = len(sys.argv) > 6
    torch.cuda.set_device(0)
    group = None
    if backend != "none":
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
        kw = dict(device_id=torch.device("cuda:0")) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    from helpers import head_cfg
    from pairnet_amd import CrossHead2, TailTrainer
    head = CrossHead2(**head_cfg())
    head.init_weights(seed=3)
    head.to("cuda:0")
    tr = TailTrainer(head, lr=1e-3, train_decoder=True, bucket_bytes=8 << 20)
    if backend == "nccl" and world == 1:        # a live RCCL communicator on the one GPU
        from pairnet_amd.dist import GradReducer
        tr.reducer = GradReducer(tr.flat_grad, bucket_bytes=8 << 20, force_collective=True)
    if noapply:
        tr.apply_gradients = lambda: None
    seed = int(os.environ.get("DDP_BATCH_SEED", 20 + rank))
    feats, metas, gt_rels, gt_labels, gt_masks, pts = batch(seed)
    tr.step(feats, metas, gt_rels, gt_labels, gt_masks, point_coords=pts)
    torch.cuda.synchronize()
    grad1 = tr.flat_grad.cpu().clone()
    if not noapply:
        tr.step(feats, metas, gt_rels, gt_labels, gt_masks, point_coords=pts)
        torch.cuda.synchronize()
    torch.save(dict(grad1=grad1, params=tr.flat_p.cpu(), scale=tr.reducer.scale,
                    collectives=tr.reducer.collectives, buckets=len(tr.reducer.bounds)), out)
    if backend != "none":
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
