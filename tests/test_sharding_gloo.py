"""N > 1 path on CPU: two processes (torch.distributed, gloo) follow bench.py's protocol --
rank = (tx+ty) % world tile ownership, compact tile-major buffers padded to a common stride,
gather to rank 0, de-interleave -- with the CPU oracle standing in for the renderer.  The
device-side equivalents (pt_render tile mode, pt_deinterleave) are covered bit-exactly on the
GPU by tests/test_gpu_parity.py::test_tile_sharding_matches_whole_frame."""
import importlib
import importlib.util
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def owned_tiles(width, height, rank, world):
    tx_n, ty_n = (width + 7) // 8, (height + 7) // 8
    return [(tx, ty) for ty in range(ty_n) for tx in range(tx_n) if (tx + ty) % world == rank]


def compact_from_full(full, rank, world, stride_floats):
    h, w = full.shape[:2]
    out = np.zeros(stride_floats, np.float32).reshape(-1, 64, 4)
    for slot, (tx, ty) in enumerate(owned_tiles(w, h, rank, world)):
        for lane in range(64):
            px, py = tx * 8 + (lane & 7), ty * 8 + (lane >> 3)
            if px < w and py < h:
                out[slot, lane] = full[py, px]
    return out.reshape(-1)


def deinterleave(gathered, width, height, world):
    full = np.zeros((height, width, 4), np.float32)
    for rank in range(world):
        buf = gathered[rank].reshape(-1, 64, 4)
        for slot, (tx, ty) in enumerate(owned_tiles(width, height, rank, world)):
            for lane in range(64):
                px, py = tx * 8 + (lane & 7), ty * 8 + (lane >> 3)
                if px < width and py < height:
                    full[py, px] = buf[slot, lane]
    return full


def _worker(rank, world, port, width, height, result_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import orc as orc_mod
    from scenes import random_soup
    rt = importlib.import_module("raytracer-public_amd")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    orc = orc_mod.load()
    tris = random_soup(400, 3)
    _, bvh4 = orc.build_bvh4(tris)
    p = orc.make_params(width, height, 400, cam_pos=(0, 0, 2.2), mode=orc_mod.MODE_PATH, spp=1, max_bounces=2, seed=9)
    full, _, _ = orc.render(p, tris, bvh4)          # every rank can compute any pixel: results are keyed by pixel, not by rank
    stride = max(rt.tile_layout(width, height, r, world)[1] for r in range(world))
    nt, fl = rt.tile_layout(width, height, rank, world)
    assert nt == len(owned_tiles(width, height, rank, world)) and fl == nt * 256
    # the order of the compact buffer is the PRODUCT's (pt_tile_ids = the list pt_render uploads): this test's own enumeration must equal it
    tx_n = (width + 7) // 8
    assert [ty * tx_n + tx for tx, ty in owned_tiles(width, height, rank, world)] == rt.tile_ids(width, height, rank, world).tolist()
    mine = torch.from_numpy(compact_from_full(full, rank, world, stride))
    glist = [torch.empty(stride) for _ in range(world)] if rank == 0 else None
    work = dist.gather(mine, glist, dst=0, async_op=True)
    work.wait()
    ok = True
    if rank == 0:
        got = deinterleave([g.numpy() for g in glist], width, height, world)
        ok = np.array_equal(got.view(np.uint32), full.view(np.uint32))
    # The same frame as PACKED shares (what bench.py and pt_group ship since round 4): only the rank's tiles inside a tile rectangle travel,
    # as 64 x (r, g, b) each, in the order of the product's pt_packed_tile_ids; everything outside the rectangle is one constant.  Here the
    # rectangle is the tight one around the pixels that differ from the camera-miss value, widened by a tile.
    tiles_x, tiles_y = (width + 7) // 8, (height + 7) // 8
    hit = np.argwhere((full[..., :3] != np.float32(0.01)).any(axis=2))
    rect = (0, 0, 0, 0)
    if hit.size:
        y0, x0 = hit.min(axis=0) // 8; y1, x1 = hit.max(axis=0) // 8 + 1
        rect = (max(int(x0) - 1, 0), max(int(y0) - 1, 0), min(int(x1) + 1, tiles_x), min(int(y1) + 1, tiles_y))
    ids = rt.packed_tile_ids(width, height, rank, world, rect)
    want_ids = [ty * tiles_x + tx for tx, ty in owned_tiles(width, height, rank, world) if rect[0] <= tx < rect[2] and rect[1] <= ty < rect[3]]
    assert ids.tolist() == want_ids
    max_tiles, pfloats = rt.packed_layout(width, height, world, rect)
    assert pfloats == max_tiles * 192 and len(want_ids) <= max_tiles
    share = np.zeros(max(pfloats, 4), np.float32)
    comp = compact_from_full(full, rank, world, stride).reshape(-1, 64, 4)
    own_ids = rt.tile_ids(width, height, rank, world).tolist()
    for slot, tid in enumerate(ids.tolist()):
        share[slot * 192:(slot + 1) * 192] = comp[own_ids.index(tid), :, :3].reshape(-1)
    plist = [torch.empty(share.size) for _ in range(world)] if rank == 0 else None
    dist.gather(torch.from_numpy(share), plist, dst=0)
    if rank == 0:
        rebuilt = np.empty((height, width, 4), np.float32); rebuilt[..., :3] = np.float32(0.01); rebuilt[..., 3] = 1.0
        for r in range(world):
            buf = plist[r].numpy()
            for slot, tid in enumerate(rt.packed_tile_ids(width, height, r, world, rect).tolist()):
                tx, ty = tid % tiles_x, tid // tiles_x
                tile = buf[slot * 192:(slot + 1) * 192].reshape(8, 8, 3)
                hh, ww = min(8, height - ty * 8), min(8, width - tx * 8)
                rebuilt[ty * 8:ty * 8 + hh, tx * 8:tx * 8 + ww, :3] = tile[:hh, :ww]
        ok = ok and np.array_equal(rebuilt.view(np.uint32), full.view(np.uint32))
        np.save(result_path, np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("width,height", [(64, 40), (70, 33)])
def test_two_rank_gather_reassembles_the_frame(tmp_path, width, height):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    result = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, port, width, height, result), nprocs=2, join=True)
    assert bool(np.load(result)[0])


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)           # only definitions run: the benchmark itself sits behind __main__
    return mod


@pytest.mark.parametrize("n_steps,batch,world", [(20, 20, 1), (256, 32, 1), (20, 20, 8), (256, 256, 8), (256, 64, 2), (7, 32, 4), (1, 1, 8), (5, 5, 8), (3, 160, 8)])
def test_bench_launch_schedule_covers_every_step_once(n_steps, batch, world):
    """bench.py's launches: every step in exactly one launch, in order, no launch larger than a batch; a sharded run is cut into
    at least SHARD_PIECES launches of about equal size (its last gather and drain are exposed, the earlier gathers travel behind the
    next launch), never growing towards the end; a single-GPU run uses full batches."""
    bench = _bench_module()
    launches = bench.schedule(n_steps, batch, world, False)
    assert launches[0][0] == 0
    done = 0
    for first, nf in launches:
        assert first == done and 1 <= nf <= batch
        done += nf
    assert done == n_steps
    sizes = [nf for _, nf in launches]
    if world == 1:
        assert all(nf == batch for nf in sizes[:-1])
    else:
        assert len(sizes) == min(n_steps, max(bench.SHARD_PIECES, -(-n_steps // batch)))
        assert all(a >= b for a, b in zip(sizes, sizes[1:])) and max(sizes) - min(sizes) <= 1
    assert bench.schedule(n_steps, batch, world, True) == [(i, min(batch, n_steps - i)) for i in range(0, n_steps, batch)]


def test_bench_schedule_for_the_drivers_own_sharded_command():
    """`--gpus 8 --steps 20 --warmup 5`: four launches of five share-frames (was 10 / 8 / 2), the warm-up 2 / 1 / 1 / 1."""
    bench = _bench_module()
    cap = min(32 * 8, 256, 20)
    assert [nf for _, nf in bench.schedule(20, cap, 8, False)] == [7, 7, 6]
    assert [nf for _, nf in bench.schedule(5, cap, 8, False)] == [2, 2, 1]
    assert [nf for _, nf in bench.schedule(256, 256, 8, False)] == [86, 85, 85]


def test_bench_submit_launches_is_the_sequence_both_callers_replay():
    """bench.py and tools/shard_schedule_sim.py submit through one function: per launch its frames with the target set first, a flush
    for a partial batch, then the caller's hook."""
    bench = _bench_module()
    calls = []

    class Ctx:
        def render(self, p): calls.append(("render", p.frame))
        def flush(self): calls.append(("flush",))

    class P: frame = 0

    bench.submit_launches(Ctx(), P(), [(0, 2), (2, 1)], 100, 2, lambda k, j, f: calls.append(("target", k, j, f)), lambda k, nf: calls.append(("after", k, nf)))
    assert calls == [("target", 0, 0, 100), ("render", 100), ("target", 0, 1, 101), ("render", 101), ("after", 0, 2),
                     ("target", 1, 0, 102), ("render", 102), ("flush",), ("after", 1, 1)]


def test_bench_busy_time_is_the_union_of_launch_intervals():
    bench = _bench_module()
    assert bench.busy_ms([0.0, 10.0], [5.0, 5.0]) == pytest.approx(10.0)          # disjoint
    assert bench.busy_ms([0.0, 2.0], [5.0, 5.0]) == pytest.approx(7.0)            # overlapping: never the sum of the spans
    assert bench.busy_ms([0.0, 1.0, 20.0], [10.0, 2.0, 1.0]) == pytest.approx(11.0)   # contained + disjoint


@pytest.mark.parametrize("width,height,world", [(64, 40, 2), (70, 33, 2), (1920, 1080, 8), (17, 9, 3), (8, 8, 5), (333, 77, 7)])
def test_tile_ids_are_the_interleaved_order_for_every_rank(width, height, world):
    """pt_tile_ids (product code, the list pt_render uploads) against the specification: row-major over the tiles with
    (tx + ty) % world == rank; pt_tile_layout's closed-form count agrees; the ranks partition the tile grid."""
    rt = importlib.import_module("raytracer-public_amd")
    tx_n, ty_n = (width + 7) // 8, (height + 7) // 8
    seen = []
    for rank in range(world):
        ids = rt.tile_ids(width, height, rank, world).tolist()
        assert ids == [ty * tx_n + tx for tx, ty in owned_tiles(width, height, rank, world)]
        nt, fl = rt.tile_layout(width, height, rank, world)
        assert nt == len(ids) and fl == nt * 256
        seen += ids
    assert sorted(seen) == list(range(tx_n * ty_n))
