#!/bin/bash
# Static look at what the compiler made of trace_paths_kernel<0, true> (no counters, short reciprocal forms: the variant every ordinary launch runs; INSTR=2 in the
# environment looks at the TIMELINE variant instead, INSTR=1 at the COUNTERS one): per loop depth of the three instances of the launch loop (one ray
# per lane / one ray per quad) the instruction count, vector instructions, register copies (v_mov_b32), idle issue slots (s_nop),
# scratch accesses and SGPR-spill lane moves.  Round 4 found 2-3 % of a dense frame in things only this shows: ~220 register copies per
# pass of the outer loop after an innocent-looking early exit, an `s_nop` behind every one-instruction asm statement, uniform flags spilled
# to scratch.  The register allocation of this kernel is fragile -- run this after any change to the loop.
#   usage (no GPU needed): tools/isa_stats.sh [extra compiler flags, e.g. -DPT_QUAD=2]
cd "$(dirname "$0")/../raytracer-public_amd/csrc" || exit 1
FL="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -enable-post-misched=false -fvisibility=hidden -I../../include $*"
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FL -S --cuda-device-only pt_megakernel.hip -o $TMP/mk.s 2>/dev/null || { echo "compile failed"; exit 1; }
awk "/^_ZN3ptk18trace_paths_kernelILi${INSTR:-0}ELb1/,/^\\.Lfunc_end/" $TMP/mk.s > $TMP/k.s      # the whole function: a variant may hold blocks behind its first s_endpgm
python3 - $TMP/k.s <<'PY'
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
# The launch loop is instantiated three times (pt_megakernel.hip).  Every basic block is attributed to its OUTERMOST loop (the instance; listed in program
# order) and, inside it, to the depth-2 loop it belongs to: an instance's outer loop body holds the shade and refill passes, its depth-2 loops are the
# camera-ray generation (one pass per 64 rays) and the TRAVERSAL STEPS -- the hot loop --, depth 3 the pop loop and the deep-stack spill loop.
names = ["one ray per lane, dense part (A)", "one ray per lane, nothing left to start (B: shadow rays to idle lanes)", "one ray per quad (C)"]
outer_of, mid_of, depth_of = {}, {}, {}
for i, l in enumerate(lines):            # loop headers: their depth, their outermost loop and their depth-2 ancestor
    if re.match(r'^\.LBB', l):
        ctx = ' '.join(lines[i:i + 6]); label = re.match(r'^\.LBB(\d+_\d+)', l).group(1)
        m = re.search(r'Loop Header: Depth=(\d+)', ctx)
        if m:
            d = int(m.group(1)); depth_of[label] = d
            p1 = re.search(r'Parent Loop BB(\d+_\d+) Depth=1', ctx); p2 = re.search(r'Parent Loop BB(\d+_\d+) Depth=2', ctx)
            outer_of[label] = p1.group(1) if p1 else label
            mid_of[label] = label if d == 2 else (p2.group(1) if p2 else None)
c = collections.defaultdict(lambda: collections.defaultdict(collections.Counter))      # instance -> (depth-2 loop or None, depth) -> counters
# SGPR-spill moves are attributed to their basic block first: a block that also touches the deep-stack spill area (a 64-bit global store without a
# scalar base, or a flat 64-bit load: stack entries beyond the LDS short stack, 0.15 % of the pushes) is COLD -- the compiler spills what only such
# blocks need, which is what one wants; `sgpr_spill_moves_hot` counts the moves in all other blocks
order, cur, blk, blk_moves, blk_cold, blk_ctx, kinds = [], None, None, collections.Counter(), set(), {}, collections.defaultdict(set)
def where(l, i, label):
    if label in depth_of: return (outer_of[label], mid_of[label], depth_of[label])
    m = re.search(r'in Loop: Header=BB(\d+_\d+) Depth=(\d+)', l)
    return (outer_of.get(m.group(1)), mid_of.get(m.group(1)), int(m.group(2))) if m else None
for i, l in enumerate(lines):
    if re.match(r'^\.LBB', l):
        label = re.match(r'^\.LBB(\d+_\d+)', l).group(1); blk = label; cur = where(l, i, label); blk_ctx[blk] = cur
        if cur and cur[0] not in order: order.append(cur[0])
    elif re.match(r'^; %bb\.', l):
        blk = 'bb' + l.split('.')[1].split(':')[0] + '@' + str(i); cur = where(l, i, None); blk_ctx[blk] = cur       # a fall-through block names its loop itself (none: outside every loop)
    t = l.strip()
    if cur is None or cur[0] is None or not re.match(r'^(v_|s_|ds_|global_|scratch_|buffer_|flat_)', t): continue
    op = t.split()[0]
    k = c[cur[0]][(cur[1], cur[2])]
    k['all'] += 1
    if op in ('v_readlane_b32', 'v_writelane_b32'): blk_moves[blk] += 1
    if (op == 'global_store_dwordx2' and t.rstrip().endswith('off')) or op == 'flat_load_dwordx2': blk_cold.add(blk)
    if op == 'global_atomic_add': kinds[cur[1]].add('claim')
    if op == 'ds_write_b64' or op == 'ds_write2_b32': kinds[cur[1]].add('push')
    for key, hit in (('valu', op.startswith('v_')), ('salu', op.startswith('s_') and op != 's_nop'), ('v_mov', op.startswith('v_mov_b32')), ('s_nop', op == 's_nop'), ('scratch', op.startswith('scratch_')),
                     ('sgpr_spill_moves', op in ('v_readlane_b32', 'v_writelane_b32')), ('branches', op.startswith('s_cbranch')), ('dpp', 'dpp' in t)):
        if hit: k[key] += 1
for b, n in blk_moves.items():
    w = blk_ctx.get(b)
    if w and w[0] is not None and b not in blk_cold: c[w[0]][(w[1], w[2])]['sgpr_spill_moves_hot'] += n
big = [o for o in order if sum(v['all'] for v in c[o].values()) > 300]
for n, o in enumerate(big):
    print(names[n] if n < len(names) and len(big) == 3 else "loop BB%s" % o)
    for (mid, d) in sorted(c[o], key=lambda kd: (kd[0] is not None, kd[0] or '', kd[1])):
        if mid is None: what = "outer loop body (shade pass, refill pass, pass conditions)"
        else:
            kind = "TRAVERSAL STEPS" if 'push' in kinds[mid] and 'claim' not in kinds[mid] else ("camera-ray generation, one pass per 64 rays" if 'claim' in kinds[mid] else "other")
            what = ("loop BB%s, depth 2: %s" % (mid, kind)) if d == 2 else ("  its depth-%d loops (pop loop, deep-stack spill loop)" % d)
        print("  %s: %s" % (what, dict(c[o][(mid, d)])))
PY
make -s resource-usage EXTRA="$*" 2>&1 | grep -A9 "trace_paths_kernelILi${INSTR:-0}ELb1" | grep -E "VGPRs:|ScratchSize|Spill" | sed "s/.*remark: *//"
rm -rf $TMP
