"""The megakernel's launch heuristics (pt_api.cpp::plan_launch) as a pure function, pinned for the launch shapes bench.py uses: a lone whole frame,
the driver's 20-frame launch, the sustained leg's 32-frame launches, launches of five 1/8 shares (the N = 8 run of earlier rounds; seven now) and the N = 2 run's of 64 halves.
No GPU: pt_debug_launch_plan needs no context.  (What the numbers were measured with: DESIGN.md section 6.1, profiles/HISTORY.md, tools/ab/sweep.sh, tools/ab/pipe_sweep.sh.)"""
import ctypes as C
import importlib

import pytest

rt = importlib.import_module("raytracer-public_amd")

CUS = 256
FULL_GRID = CUS * 6 * 4            # six single-wavefront workgroups per SIMD
C2_TILES, C2_TRACED, SPP = 240 * 135, 97 * 92, 4      # 1920 x 1080 in 8 x 8 tiles; a dragon-class rectangle of traced tiles (its size does not enter the heuristics)
NAMES = ("grid", "perm_rows", "perm_cols", "total_items", "chunk_items", "xcd_span", "shade_threshold", "fill_threshold", "quad_live", "fork_shadow", "slots", "setup_slots")


def plan(frames, tile_count=1, in_flight=0, traced_tiles=C2_TRACED, batch_size=None):
    out = (C.c_uint32 * 12)()
    batches = frames * (traced_tiles // tile_count) * SPP
    rc = rt.lib.pt_debug_launch_plan(C.c_uint32(CUS), C.c_uint32(frames), C.c_uint32(tile_count), C.c_uint32(in_flight), C.c_uint32(batches), C.c_uint32(batch_size or frames), out)
    assert rc == 0, rt.lib.pt_last_error(None)
    d = dict(zip(NAMES, out))
    d["batches"] = batches
    return d


def check_common(p):
    assert p["chunk_items"] == 128                                   # two batches of 64 per claim
    assert (p["shade_threshold"], p["fill_threshold"], p["quad_live"], p["fork_shadow"]) == (16, 4, 16, 2)
    assert p["total_items"] == p["perm_rows"] * p["perm_cols"] * 64
    assert p["batches"] <= p["perm_rows"] * p["perm_cols"] < p["batches"] + p["perm_rows"]      # the transposition covers every batch, with less than one row of padding
    assert p["setup_slots"] >= p["slots"] >= 1
    if p["xcd_span"]:
        assert p["xcd_span"] % p["chunk_items"] == 0 and 8 * p["xcd_span"] >= p["total_items"]


def test_lone_whole_frame_and_one_render_per_frame():
    p = plan(1)
    check_common(p)
    assert p["grid"] == FULL_GRID                                    # a lone frame keeps the whole grid ...
    assert p["perm_rows"] == 128 and p["xcd_span"] == 0              # ... cuts the frame into 128 segments (balance), one queue
    assert p["slots"] == 6
    for in_flight, div in ((1, 2), (2, 3), (3, 4), (5, 4)):          # ... and shrinks with the launches already in flight (the reference's call shape without waits)
        assert plan(1, in_flight=in_flight)["grid"] == -(-FULL_GRID // div)


def test_the_drivers_twenty_frame_launch():
    p = plan(20)
    check_common(p)
    assert p["grid"] == FULL_GRID
    assert p["perm_rows"] == 20                                      # one row per frame: all frames walk the image in step
    assert p["xcd_span"] != 0                                        # XCD-aware queue from 8 frames of work on
    assert p["slots"] == 4


def test_sustained_thirty_two_frame_launches():
    p = plan(32)
    check_common(p)
    assert p["grid"] == FULL_GRID and p["perm_rows"] == 32 and p["xcd_span"] != 0
    assert p["slots"] == 3                                           # a long launch only needs its tail covered by the next one


def test_five_eighth_shares_per_launch():
    p = plan(5, tile_count=8)
    check_common(p)
    assert p["grid"] == FULL_GRID                                    # grid / 4 for ONE 1/8 share, times five shares: the whole grid again
    assert plan(1, tile_count=8)["grid"] == FULL_GRID // 4
    assert p["perm_rows"] == 64 * 5 and p["xcd_span"] == 0           # 5/8 of a frame of work: a short launch
    assert p["slots"] == 8                                           # small sharded launches need several in flight to fill the chip
    assert plan(5, tile_count=8, batch_size=20)["setup_slots"] == 8  # the slots are set up for a full batch of the current setting as well


def test_sixty_four_halves_per_launch():
    p = plan(64, tile_count=2)
    check_common(p)
    assert p["grid"] == FULL_GRID and p["perm_rows"] == 64 and p["xcd_span"] != 0
    assert p["slots"] == 3


def test_mid_size_launches_take_one_row_per_frame_from_three_frames_of_work():
    # (profiles/r05_m2_midsize_rows.txt: 3 .. 7 frames of work 1 .. 7 % faster than with 128 segments per frame; the XCD-aware queue still starts at 8)
    for frames, tile_count, rows in ((2, 1, 128 * 2), (3, 1, 3), (5, 1, 5), (7, 1, 7), (5, 2, 128 * 5), (6, 2, 6), (10, 2, 10), (16, 4, 16), (20, 8, 128 * 20), (24, 8, 24)):
        p = plan(frames, tile_count=tile_count)
        check_common(p)
        assert p["perm_rows"] == rows, (frames, tile_count, p["perm_rows"])
        assert (p["xcd_span"] != 0) == (frames * 8 // tile_count >= 64)


def test_plan_invariants_over_random_launch_shapes():
    # whatever the shape, the plan covers every batch exactly once (plus less than a row of padding), keeps chunks whole batches, never leaves the grid empty and
    # never lets the 32-bit queue cursor wrap
    import random
    rnd = random.Random(5)
    for _ in range(400):
        tile_count = rnd.choice((1, 1, 1, 2, 3, 4, 8, 16))
        frames = rnd.choice((1, 2, 3, 5, 8, 13, 20, 32, 64, 100, 256))
        traced = rnd.randint(1, C2_TILES)
        p = plan(frames, tile_count=tile_count, in_flight=rnd.randint(0, 6), traced_tiles=max(traced, tile_count), batch_size=rnd.choice((None, 256)))
        check_common(p)
        assert 1 <= p["grid"] <= FULL_GRID
        assert p["chunk_items"] % 64 == 0 and p["total_items"] % 64 == 0
        assert 1 <= p["perm_rows"] <= 4096
        assert p["total_items"] + (p["grid"] + 1) * p["chunk_items"] <= 0xFFFFFFFF
        assert 1 <= p["slots"] <= 8


def test_a_launch_too_large_for_the_queue_cursor_is_refused():
    out = (C.c_uint32 * 12)()
    rc = rt.lib.pt_debug_launch_plan(C.c_uint32(CUS), C.c_uint32(256), C.c_uint32(1), C.c_uint32(0), C.c_uint32(0x03FFFFF0), C.c_uint32(256), out)
    assert rc != 0
