"""What the compiler made of the megakernel, pinned (compile-only: hipcc cross-compiles gfx950 without a GPU).

The kernel sits at exactly 80 vector registers -- six wavefronts per SIMD -- with no scratch memory, and its register allocation is fragile
(tools/isa_stats.sh): an innocent edit of the launch loop can cost a spill or a wavefront of occupancy without failing any parity test.
The second test guards the tree against the one inline-asm pattern that hung a GPU in round 4: a memory LOAD written as an asm statement without an
output operand -- the compiler then considers the destination register free while the load is still in flight (tools/README.md)."""
import glob
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "raytracer-public_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")


FLAGS = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -enable-post-misched=false -fvisibility=hidden".split()


@pytest.fixture(scope="module")
def megakernel_asm(tmp_path_factory):
    """The device assembly of pt_megakernel.hip with the library's flags (compile-only, ~15 s), once per module."""
    asm = tmp_path_factory.mktemp("mk") / "mk.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", *FLAGS, "-I" + os.path.join(os.path.dirname(HERE), "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, "pt_megakernel.hip"), "-o", str(asm)], check=True, capture_output=True, timeout=900)
    return asm.read_text()


def kernel_body(text, instr, bounded):
    """One variant's whole function (a variant may hold blocks behind its first s_endpgm)."""
    start = text.index("_ZN3ptk18trace_paths_kernelILi%dELb%dEEEvNS_10RenderArgsE:" % (instr, bounded))
    return text[start:text.index(".Lfunc_end", start)]


@pytest.mark.skipif(HIPCC is None, reason="hipcc is missing")
def test_megakernel_registers_scratch_and_occupancy_are_pinned():
    out = subprocess.run(["make", "-s", "-C", CSRC, "resource-usage"], capture_output=True, text=True, timeout=600)
    text = out.stdout + out.stderr
    blocks = re.split(r"remark: Function Name: ", text)
    seen = {}
    for b in blocks[1:]:
        name = b.split()[0]
        m = re.match(r"_ZN3ptk18trace_paths_kernelILi([012])ELb([01])E", name)
        if not m:
            continue
        f = {k: int(v) for k, v in re.findall(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", b)}
        seen[(int(m.group(1)), int(m.group(2)))] = f
    assert set(seen) == {(i, b) for i in (0, 1, 2) for b in (0, 1)}, sorted(seen)     # <INSTR, BOUNDED>: production, COUNTERS, TIMELINE x both reciprocal forms
    for bounded in (0, 1):                                                        # the production variants (no counters)
        f = seen[(0, bounded)]
        assert f["VGPRs"] <= 80, f
        assert f["ScratchSize [bytes/lane]"] == 0, f
        assert f["VGPRs Spill"] == 0, f
        assert f["Occupancy [waves/SIMD]"] >= 6, f
        assert f["SGPRs Spill"] <= 18, f                                           # 17: the kernel-argument pointer, the workgroup id, launch constants -- none of them moved inside the traversal steps (next test)
        assert f["LDS Size [bytes/block]"] <= 160 * 1024 // 24, f                  # six single-wave workgroups per SIMD fit the CU's LDS
    for bounded in (0, 1):
        # the TIMELINE variant (tools/wave_timeline.py): production registers, production occupancy, NO scratch -- its bookkeeping is wave-uniform (scalar
        # registers, events written when they happen), so what it times is the production kernel.  (The COUNTERS variant spills ~250 registers: never time with it.)
        f = seen[(2, bounded)]
        assert f["VGPRs"] <= 80 and f["ScratchSize [bytes/lane]"] == 0 and f["VGPRs Spill"] == 0 and f["Occupancy [waves/SIMD]"] >= 6, f
        assert f["SGPRs Spill"] <= 48, f
        assert seen[(1, bounded)]["ScratchSize [bytes/lane]"] > 0                   # if this ever becomes 0 the two variants can be merged again


def loop_census(body):
    """Per depth-2 loop of a kernel body: what it is and how many SGPR-spill moves (v_readlane / v_writelane) it executes outside the blocks that touch the
    deep-stack spill area.  The logic of tools/isa_stats.sh: loop headers carry their depth and parents in the compiler's comments; every basic block names the
    loop it is in; the TRAVERSAL STEPS are the depth-2 loop that pushes to the LDS stack (ds_write_b64) and does not claim queue items (global_atomic_add);
    a block with a 64-bit global store without a scalar base or a 64-bit flat load is the deep-stack spill path (0.15 % of the pushes): cold."""
    lines = body.split("\n")
    depth_of, mid_of = {}, {}
    for i, l in enumerate(lines):
        m = re.match(r"^\.LBB(\d+_\d+)", l)
        if not m:
            continue
        ctx = " ".join(lines[i:i + 6])
        d = re.search(r"Loop Header: Depth=(\d+)", ctx)
        if d:
            label = m.group(1); depth_of[label] = int(d.group(1))
            p2 = re.search(r"Parent Loop BB(\d+_\d+) Depth=2", ctx)
            mid_of[label] = label if depth_of[label] == 2 else (p2.group(1) if p2 else None)
    loops = {}
    cur, blk = None, None
    blocks = {}
    for i, l in enumerate(lines):
        m = re.match(r"^\.LBB(\d+_\d+)", l)
        if m or re.match(r"^; %bb\.", l):
            blk = i
            if m and m.group(1) in depth_of:
                cur = mid_of[m.group(1)] if depth_of[m.group(1)] >= 2 else None
            else:
                h = re.search(r"in Loop: Header=BB(\d+_\d+) Depth=(\d+)", l)
                cur = mid_of.get(h.group(1)) if h and int(h.group(2)) >= 2 else None
            blocks[blk] = {"loop": cur, "moves": 0, "cold": False}
        t = l.strip()
        if blk is None or not re.match(r"^(v_|s_|ds_|global_|scratch_|buffer_|flat_)", t):
            continue
        op = t.split()[0]
        b = blocks[blk]
        if op in ("v_readlane_b32", "v_writelane_b32"):
            b["moves"] += 1
        if (op == "global_store_dwordx2" and t.rstrip().endswith("off")) or op == "flat_load_dwordx2":
            b["cold"] = True
        if b["loop"] is not None:
            k = loops.setdefault(b["loop"], {"push": False, "claim": False, "valu": 0, "all": 0})
            k["all"] += 1
            k["valu"] += op.startswith("v_")
            k["push"] |= op in ("ds_write_b64", "ds_write2_b32")
            k["claim"] |= op == "global_atomic_add"
    for b in blocks.values():
        if b["loop"] is not None:
            loops[b["loop"]]["hot_moves"] = loops[b["loop"]].get("hot_moves", 0) + (0 if b["cold"] else b["moves"])
    return loops


@pytest.mark.skipif(HIPCC is None, reason="hipcc is missing")
def test_no_sgpr_spill_move_inside_the_traversal_steps(megakernel_asm):
    """The property that matters about the 17 spilled SGPRs: NONE of them is read or written inside the traversal steps of the production variants -- the loop
    that runs 3.5 M times per frame and wavefront-step.  (The moves it contains sit in the blocks of the deep-stack spill path.)"""
    for bounded in (0, 1):
        loops = loop_census(kernel_body(megakernel_asm, 0, bounded))
        steps = [k for k in loops.values() if k["push"] and not k["claim"]]
        assert len(steps) == 3, (bounded, loops)                                 # the loop's three instances: dense, shadow rays to idle lanes, one ray per quad
        for k in steps:
            assert k["hot_moves"] == 0, (bounded, k)
        dense = steps[0]
        assert 440 <= dense["all"] <= 500 and dense["valu"] <= 265, dense         # 488 instructions, 262 of them vector, incl. the pop loop and the deep-stack blocks (tools/isa_stats.sh lists those apart: 441 / 239 + 37 / 15 + ...): a regression of the step shows here first


@pytest.mark.skipif(HIPCC is None, reason="hipcc is missing")
def test_the_step_waits_for_the_pieces_of_its_record_one_by_one(megakernel_asm):
    """Round 5: the traversal step asks for the four 16-byte pieces of its record at once and waits for them piece by piece (node side first: child 3, 2, 1, 0).
    That is a property of what the compiler emits -- load order and `s_waitcnt vmcnt(3..0)` placement --, not of the source alone: pinned here for the production
    variant's two one-ray-per-lane instances of the loop."""
    body = kernel_body(megakernel_asm, 0, 1)
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    groups = 0
    for i in range(len(lines) - 4):
        four = lines[i:i + 4]
        if not all(l.startswith("global_load_dwordx4") for l in four):
            continue
        if [re.search(r"offset:(\d+)", l).group(1) if "offset:" in l else "0" for l in four] != ["48", "32", "16", "0"]:
            continue
        waits = []
        for l in lines[i + 4:]:
            if l.startswith("global_load") or l.startswith("global_store") or l.startswith("s_cbranch_execz .LBB") and len(waits) >= 4:
                break
            m = re.match(r"s_waitcnt vmcnt\((\d)\)$", l)
            if m:
                waits.append(int(m.group(1)))
                if len(waits) == 4:
                    break
        assert waits == [3, 2, 1, 0], (i, waits)
        groups += 1
    assert groups == 2, groups            # the dense instance and the one that hands shadow rays to idle lanes


LOADS = re.compile(r"\b(ds_read\w*|ds_load\w*|global_load\w*|buffer_load\w*|flat_load\w*|scratch_load\w*|s_load\w*|s_buffer_load\w*|global_atomic\w*|ds_\w*_rtn\w*)\b")


def _asm_statements(src):
    """(line, string part, [operand sections]) of every asm statement: text up to the matching parenthesis, split at top-level colons outside strings."""
    for m in re.finditer(r"\basm\s*(volatile)?\s*\(", src):
        i, depth, in_str, sections, cur = m.end(), 1, False, [], []
        while i < len(src) and depth:
            ch = src[i]
            if in_str:
                cur.append(ch)
                if ch == "\\":
                    cur.append(src[i + 1]); i += 1
                elif ch == '"':
                    in_str = False
            elif ch == '"':
                in_str = True; cur.append(ch)
            elif ch == "(":
                depth += 1; cur.append(ch)
            elif ch == ")":
                depth -= 1
                if depth:
                    cur.append(ch)
            elif ch == ":" and depth == 1 and src[i + 1] != ":" and src[i - 1] != ":":
                sections.append("".join(cur)); cur = []
            else:
                cur.append(ch)
            i += 1
        sections.append("".join(cur))
        yield src.count("\n", 0, m.start()) + 1, sections[0], sections[1:]


def test_no_asm_memory_load_without_an_output_operand():
    checked = 0
    for path in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(CSRC, "*.cpp"))):
        src = open(path, encoding="utf-8", errors="replace").read()
        for line, text, ops in _asm_statements(src):
            if not LOADS.search(text):
                continue
            checked += 1
            outputs = ops[0].strip() if ops else ""
            assert outputs, "%s:%d: an asm statement loads from memory but declares no output operand: the destination register is dead to the compiler while the load is in flight" % (os.path.basename(path), line)
            # the loaded registers must not be handed out to the inputs either (early clobber), and the data must have arrived when the statement ends
            for dst in re.findall(r'"(=[^"]*)"', outputs):
                if "v" in dst:
                    assert "&" in dst, "%s:%d: a loaded vector register is not an early-clobber output (\"=&v\")" % (os.path.basename(path), line)
            assert re.search(r"s_waitcnt\s+(lgkmcnt|vmcnt)\(0\)", text), "%s:%d: the asm statement does not wait for its own load" % (os.path.basename(path), line)
    assert checked >= 1          # the hand-written pop loop (pt_megakernel_loop.inc) is such a statement


@pytest.mark.skipif(HIPCC is None, reason="hipcc is missing")
def test_step_census_matches_bench_constants(megakernel_asm):
    """bench.py's `lane_utilisation_by_side` prices the lanes of a step's sides with static vector-instruction counts (bench.STEP_VALU).  Those are a property of
    what the compiler emits for THIS source: re-derived here from the dense instance of the production variant -- the block with the twelve v_perm_b32 is the
    node side's four slab tests, the block with the v_rcp_f32 the leaf side -- and compared (+- 8 instructions)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.path.dirname(HERE), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    body = kernel_body(megakernel_asm, 0, 1).split("\n")
    blocks, cur = [], None
    for l in body:
        if re.match(r"^\.LBB", l) or re.match(r"^; %bb\.", l):
            cur = {"valu": 0, "perm": 0, "rcp": 0, "stores": 0, "depth2": "Depth=2" in l}
            blocks.append(cur)
        t = l.strip()
        if cur is None:
            continue
        cur["valu"] += t.startswith("v_")
        cur["perm"] += t.startswith("v_perm_b32")
        cur["rcp"] += t.startswith("v_rcp_f32")
        cur["stores"] += t.startswith("ds_write_b64")
    slab = next(i for i, b in enumerate(blocks) if b["perm"] == 12)                 # first instance = the dense one
    order = blocks[slab + 1]                                                        # child order + stack words (executed when a child is hit)
    stores = next(b for b in blocks[slab + 1: slab + 6] if b["stores"] == 3)        # the three unconditional LDS stores
    leaf = next(b for b in blocks[slab:] if b["rcp"] == 1 and b["valu"] > 40)       # Moller-Trumbore
    node_side = blocks[slab]["valu"] + order["valu"] + stores["valu"]
    assert abs(node_side - bench.STEP_VALU["node_side"]) <= 8, (node_side, bench.STEP_VALU)
    assert abs(leaf["valu"] + 3 - bench.STEP_VALU["leaf_side"]) <= 8, (leaf["valu"], bench.STEP_VALU)
