"""RCCL with one rank on one GPU: communicator set-up and the bench's collectives
(all_gather_into_tensor of the triplet records on a side stream, all_reduce(MAX) of the
timing scalar, barrier) through the real library -- what a 1-GPU box can execute of the
multi-GPU path."""
import os, sys, time, torch
import torch.distributed as dist
import json, socket
_s = socket.socket(); _s.bind(("127.0.0.1", 0)); _port = _s.getsockname()[1]; _s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_port)
for _k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
    os.environ.pop(_k, None)
torch.cuda.set_device(0)
t = time.perf_counter()
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
send = torch.arange(2 * 6100, dtype=torch.float32, device=dev).view(2, 6100)
recv = torch.empty_like(send)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    dist.all_gather_into_tensor(recv, send)
side.synchronize()
x = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
ok = bool(torch.equal(recv, send)) and float(x) == 1.25
t_init = time.perf_counter() - t
print("backend", dist.get_backend(), "ok", ok, "%.2f s" % t_init)
n = 200
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(n):
    dist.all_gather_into_tensor(recv, send)
torch.cuda.synchronize()
us = 1e6 * (time.perf_counter() - t) / n
print("all_gather_into_tensor of 2 x 24.4 KB records, 1 rank: %.1f us per call" % us)
print("RCCL_JSON " + json.dumps({"backend": dist.get_backend(), "ranks": 1, "results_correct": ok,
                                 "init_s": t_init, "all_gather_into_tensor_us": us,
                                 "record_bytes": 2 * 6100 * 4}))
dist.destroy_process_group()
