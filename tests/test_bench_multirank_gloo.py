"""bench.py's collective protocol with MORE ranks than a one-GPU box can host, on the CPU (gloo): the GatherPipeline class bench.py itself drives
(ship / finish per launch, two buffer slots, rank 0 consuming launch b-1 behind launch b) over the launch schedule of the DRIVER'S OWN 8-GPU command
(`--gpus 8 --steps 20 --warmup 5`: warm-up [2, 2, 1], timed [7, 7, 6]) with eight processes -- every rank's share of every frame of every launch must
arrive in rank 0's buffers at [rank][frame] -- and what happens when a rank dies in the middle: the survivors leave a collective with an error within
the timeout and the job exits non-zero (nobody waits for the driver's 600 s limit).  No rendering here: the shares are seeded patterns."""
import importlib.util
import os
import socket
import sys
import time

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _share(rank, frame, floats):
    """The pattern rank `rank` ships for global frame index `frame`."""
    return (np.arange(floats, dtype=np.float32) * np.float32(0.5) + np.float32(rank * 1000 + frame)).astype(np.float32)


def _worker(rank, world, port, steps, warmup, die, result_dir):
    sys.path.insert(0, ROOT)
    os.environ["PT_BENCH_TIMEOUT"] = "20"
    import torch
    import torch.distributed as dist
    bench = _bench()
    bench.init_distributed(dist, "gloo", rank, world, init_method="tcp://127.0.0.1:%d" % port)
    try:
        floats = 48
        cap = max(1, min(32 * world, 256, steps))
        batch = max([nf for _, nf in bench.schedule(steps, cap, world, False)] + [nf for _, nf in bench.schedule(max(warmup, 1), cap, world, False)])
        packed = [torch.zeros(batch, floats) for _ in range(2)]
        gathered = [torch.zeros(world, batch, floats) for _ in range(2)] if rank == 0 else [None, None]
        seen = []                                  # rank 0: (first frame, nf, ok) per finished launch
        frames_of_slot = {}

        def on_finish(slot, nf):
            first = frames_of_slot[slot]
            ok = all(np.array_equal(gathered[slot][r][j].numpy(), _share(r, first + j, floats)) for r in range(world) for j in range(nf))
            seen.append((first, nf, ok))

        pipe = bench.GatherPipeline(dist, torch, rank, world, packed, gathered, on_finish, host_stage=False)
        launches_done = 0
        for tag, n_steps, first_frame in (("warmup", warmup, steps), ("timed", steps, 0)):
            dist.barrier()
            for k, (first, nf) in enumerate(bench.schedule(n_steps, batch, world, False)):
                if die is not None and rank == die[0] and tag == "timed" and k == die[1]:
                    raise RuntimeError("injected failure on rank %d" % rank)
                slot = k & 1
                for j in range(nf):
                    packed[slot][j].copy_(torch.from_numpy(_share(rank, first_frame + first + j, floats)))
                frames_of_slot[slot] = first_frame + first
                pipe.ship(slot, nf)
                launches_done += 1
            pipe.finish()
            dist.barrier()
        if rank == 0:
            np.save(os.path.join(result_dir, "seen.npy"), np.array(seen, np.int64))
            np.save(os.path.join(result_dir, "gathers.npy"), np.array([pipe.gathers]))
        dist.destroy_process_group()
    except BaseException as exc:
        open(os.path.join(result_dir, "failed_rank%d.txt" % rank), "w").write("%s: %s" % (type(exc).__name__, exc))
        bench.abort_rank(dist, rank, exc, code=4)


def _spawn(world, steps, warmup, die, tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.spawn(_worker, args=(world, port, steps, warmup, die, str(tmp_path)), nprocs=world, join=False)
    t0 = time.time()
    while any(p.is_alive() for p in ctx.processes) and time.time() - t0 < 120:
        time.sleep(0.2)
    alive = [p.is_alive() for p in ctx.processes]
    for p in ctx.processes:
        if p.is_alive():
            p.kill()                                   # by pid: a rank that outlived the timeout is the failure this test exists for
    return [p.exitcode for p in ctx.processes], alive, time.time() - t0


def test_eight_rank_rehearsal_of_the_drivers_command(tmp_path):
    """`--gpus 8 --steps 20 --warmup 5`: schedule [7, 7, 6] after a [2, 2, 1] warm-up, eight ranks, every share of every frame lands at
    gathered[slot][rank][frame] before rank 0 consumes the launch; six gathers in all."""
    bench = _bench()
    cap = min(32 * 8, 256, 20)
    assert [nf for _, nf in bench.schedule(20, cap, 8, False)] == [7, 7, 6]
    codes, alive, dt = _spawn(8, 20, 5, None, tmp_path)
    assert not any(alive) and codes == [0] * 8, (codes, alive)
    seen = np.load(str(tmp_path / "seen.npy"))
    assert seen.tolist() == [[20, 2, 1], [22, 2, 1], [24, 1, 1], [0, 7, 1], [7, 7, 1], [14, 6, 1]]
    assert int(np.load(str(tmp_path / "gathers.npy"))[0]) == 6


@pytest.mark.parametrize("die", [(2, 1), (0, 2)])
def test_a_rank_that_dies_mid_run_takes_the_job_down_within_the_timeout(tmp_path, die):
    """Rank 2 (a sender) or rank 0 (the root of every gather) raises before its second / third timed launch: it reports, tears its group down
    best-effort and exits 4 (bench.abort_rank); every other rank leaves its next collective with an error -- a closed connection at once or the
    20 s collective timeout of this test (120 s in bench.py, PT_BENCH_TIMEOUT) -- and exits non-zero too.  Nobody is still alive after 120 s."""
    codes, alive, dt = _spawn(4, 20, 5, die, tmp_path)
    assert not any(alive), "ranks still waiting in a collective after %.0f s: %s" % (dt, alive)
    assert codes[die[0]] == 4
    assert all(c not in (0, None) for c in codes), codes
    assert dt < 90
    assert "injected failure" in open(str(tmp_path / ("failed_rank%d.txt" % die[0]))).read()


def test_collectives_carry_a_timeout_and_failures_exit_nonzero():
    """The properties the first real 8-GPU run depends on, read off bench.py itself: the process group is created with a timeout well under the
    driver's 600 s limit, a failing rank goes through abort_rank, and nothing in the failure path re-executes the process."""
    bench = _bench()
    assert 10 <= bench.COLLECTIVE_TIMEOUT_S <= 300
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "timeout=timedelta(seconds=COLLECTIVE_TIMEOUT_S)" in src
    assert "abort_rank(dist, rank, exc)" in src
    assert "os.exec" not in src and "execv" not in src
