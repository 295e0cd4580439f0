#!/bin/bash
# Static look at what the compiler made of trace_paths_kernel<false, true> (no counters, short reciprocal forms: the variant every ordinary launch runs): per loop depth of the three instances of the launch loop (one ray
# per lane / one ray per quad) the instruction count, vector instructions, register copies (v_mov_b32), idle issue slots (s_nop),
# scratch accesses and SGPR-spill lane moves.  Round 4 found 2-3 % of a dense frame in things only this shows: ~220 register copies per
# pass of the outer loop after an innocent-looking early exit, an `s_nop` behind every one-instruction asm statement, uniform flags spilled
# to scratch.  The register allocation of this kernel is fragile -- run this after any change to the loop.
#   usage (no GPU needed): tools/isa_stats.sh [extra compiler flags, e.g. -DPT_QUAD=2]
cd "$(dirname "$0")/../raytracer-public_amd/csrc" || exit 1
FL="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -enable-post-misched=false -fvisibility=hidden -I../../include $*"
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FL -S --cuda-device-only pt_megakernel.hip -o $TMP/mk.s 2>/dev/null || { echo "compile failed"; exit 1; }
awk '/^_ZN3ptk18trace_paths_kernelILb0ELb1/,/s_endpgm/' $TMP/mk.s > $TMP/k.s
python3 - $TMP/k.s <<'PY'
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
# the launch loop is instantiated three times (pt_megakernel.hip): every basic block is attributed to its OUTERMOST loop, the loops are listed in program order
names = ["one ray per lane, dense part (A)", "one ray per lane, nothing left to start (B: shadow rays to idle lanes)", "one ray per quad (C)"]
outer_of, order, cur_outer, cur_depth = {}, [], None, 0
for i, l in enumerate(lines):            # loop headers: which outermost loop they belong to
    if re.match(r'^\.LBB', l):
        ctx = ' '.join(lines[i:i + 5]); label = re.match(r'^\.LBB(\d+_\d+)', l).group(1)
        if 'Loop Header: Depth=' in ctx:
            m = re.search(r'Parent Loop BB(\d+_\d+) Depth=1', ctx)
            outer_of[label] = m.group(1) if m else label
c = collections.defaultdict(lambda: collections.defaultdict(collections.Counter))
for i, l in enumerate(lines):
    if re.match(r'^\.LBB', l):
        ctx = ' '.join(lines[i:i + 5]); label = re.match(r'^\.LBB(\d+_\d+)', l).group(1)
        if label in outer_of:
            cur_outer = outer_of[label]; cur_depth = int(re.search(r'Loop Header: Depth=(\d+)', ctx).group(1))
        else:
            m = re.search(r'in Loop: Header=BB(\d+_\d+) Depth=(\d+)', l)
            cur_outer, cur_depth = (outer_of.get(m.group(1)), int(m.group(2))) if m else (None, 0)
        if cur_outer and cur_outer not in order: order.append(cur_outer)
    t = l.strip()
    if cur_outer is None or not re.match(r'^(v_|s_|ds_|global_|scratch_|buffer_|flat_)', t): continue
    op = t.split()[0]
    k = c[cur_outer][cur_depth]
    k['all'] += 1
    for key, hit in (('valu', op.startswith('v_')), ('v_mov', op.startswith('v_mov_b32')), ('s_nop', op == 's_nop'), ('scratch', op.startswith('scratch_')),
                     ('sgpr_spill_moves', op in ('v_readlane_b32', 'v_writelane_b32')), ('branches', op.startswith('s_cbranch')), ('dpp', 'dpp' in t)):
        if hit: k[key] += 1
big = [o for o in order if sum(v['all'] for v in c[o].values()) > 300]
for n, o in enumerate(big):
    print(names[n] if n < len(names) and len(big) == 3 else "loop BB%s" % o)
    for d in sorted(c[o]): print("  loop depth %d: %s" % (d, dict(c[o][d])))
PY
make -s resource-usage EXTRA="$*" 2>&1 | grep -A9 "trace_paths_kernelILb0ELb1" | grep -E "VGPRs:|ScratchSize|Spill" | sed "s/.*remark: *//"
rm -rf $TMP
