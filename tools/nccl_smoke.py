#!/usr/bin/env python3
"""RCCL mechanics used by bench.py at N > 1, exercised with world_size 1 on one GPU: gather of a batched compact
buffer issued on the context's own stream (torch ExternalStream), async work + wait, de-interleave after it."""
import importlib, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import numpy as np
rt = importlib.import_module("raytracer-public_amd")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
ctx = rt.Context(0)
stream = torch.cuda.ExternalStream(ctx.get_stream(), device=0)
ctx.set_triangles(rt.procedural_scene(0, 20000)); ctx.build_bvh()
w, h, batch, world = 320, 200, 4, 1
stride = rt.tile_layout(w, h, 0, 2)[1]
compact = torch.zeros(batch, stride, device="cuda"); gathered = torch.zeros(world, batch, stride, device="cuda")
ctx.set_batch(batch)
with torch.cuda.stream(stream):
    for j in range(batch):
        ctx.set_compact_buffer(compact[j].data_ptr(), stride)
        ctx.render(ctx.make_params(w, h, mode=rt.PT_MODE_PATH, spp=2, max_bounces=3, frame=j, tile_rank=0, tile_count=2))
    work = dist.gather(compact, [gathered[0]], dst=0, async_op=True)
    work.wait()
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
dist.barrier()
assert torch.equal(gathered[0], compact) and float(compact.abs().sum()) > 0 and float(t.item()) == 1.5
print("nccl smoke ok: gathered %d floats per frame x %d frames" % (stride, batch))
dist.destroy_process_group()
