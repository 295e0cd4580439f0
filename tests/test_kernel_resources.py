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


@pytest.mark.skipif(HIPCC is None, reason="hipcc is missing")
def test_megakernel_registers_scratch_and_occupancy_are_pinned():
    out = subprocess.run(["make", "-s", "-C", CSRC, "resource-usage"], capture_output=True, text=True, timeout=600)
    text = out.stdout + out.stderr
    blocks = re.split(r"remark: Function Name: ", text)
    seen = {}
    for b in blocks[1:]:
        name = b.split()[0]
        m = re.match(r"_ZN3ptk18trace_paths_kernelILb([01])ELb([01])E", name)
        if not m:
            continue
        f = {k: int(v) for k, v in re.findall(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", b)}
        seen[(int(m.group(1)), int(m.group(2)))] = f
    assert set(seen) == {(0, 0), (0, 1), (1, 0), (1, 1)}, sorted(seen)            # <STATS, BOUNDED>: all four variants are compiled
    for bounded in (0, 1):                                                        # the production variants (no counters)
        f = seen[(0, bounded)]
        assert f["VGPRs"] <= 80, f
        assert f["ScratchSize [bytes/lane]"] == 0, f
        assert f["VGPRs Spill"] == 0, f
        assert f["Occupancy [waves/SIMD]"] >= 6, f
        assert f["SGPRs Spill"] <= 24, f                                           # 15-21 over round 5, all of them in the cold paths (deep-stack spill area, launch prologue)
        assert f["LDS Size [bytes/block]"] <= 160 * 1024 // 24, f                  # six single-wave workgroups per SIMD fit the CU's LDS


@pytest.mark.skipif(HIPCC is None, reason="hipcc is missing")
def test_the_step_waits_for_the_pieces_of_its_record_one_by_one(tmp_path):
    """Round 5: the traversal step asks for the four 16-byte pieces of its record at once and waits for them piece by piece (node side first: child 3, 2, 1, 0).
    That is a property of what the compiler emits -- load order and `s_waitcnt vmcnt(3..0)` placement --, not of the source alone: pinned here for the production
    variant's two one-ray-per-lane instances of the loop."""
    flags = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -enable-post-misched=false -fvisibility=hidden".split()
    asm = tmp_path / "mk.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", *flags, "-I" + os.path.join(os.path.dirname(HERE), "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, "pt_megakernel.hip"), "-o", str(asm)], check=True, capture_output=True, timeout=900)
    text = asm.read_text()
    start = text.index("_ZN3ptk18trace_paths_kernelILb0ELb1EEEvNS_10RenderArgsE:")
    body = text[start:text.index("s_endpgm", start)]
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
