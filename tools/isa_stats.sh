#!/bin/bash
# Static look at what the compiler made of trace_paths_kernel<false>: per loop depth of the two instances of the launch loop (one ray
# per lane / one ray per quad) the instruction count, vector instructions, register copies (v_mov_b32), idle issue slots (s_nop),
# scratch accesses and SGPR-spill lane moves.  Round 4 found 2-3 % of a dense frame in things only this shows: ~220 register copies per
# pass of the outer loop after an innocent-looking early exit, an `s_nop` behind every one-instruction asm statement, uniform flags spilled
# to scratch.  The register allocation of this kernel is fragile -- run this after any change to the loop.
#   usage (no GPU needed): tools/isa_stats.sh [extra compiler flags, e.g. -DPT_QUAD=2]
cd "$(dirname "$0")/../raytracer-public_amd/csrc" || exit 1
FL="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -enable-post-misched=false -fvisibility=hidden -I../../include $*"
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FL -S --cuda-device-only pt_megakernel.hip -o $TMP/mk.s 2>/dev/null || { echo "compile failed"; exit 1; }
awk '/^_ZN3ptk18trace_paths_kernelILb0/,/s_endpgm/' $TMP/mk.s > $TMP/k.s
python3 - $TMP/k.s <<'PY'
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
split = next((i for i, l in enumerate(lines) if 'ds_bpermute' in l), len(lines))      # the re-seating sits between the two loop instances
for name, part in (("one ray per lane", lines[:split]), ("one ray per quad", lines[split:])):
    cur, c = 0, collections.defaultdict(collections.Counter)
    for l in part:
        if re.match(r'^\.LBB|^; %bb', l):
            m = re.search(r'Depth=(\d+)', l); cur = int(m.group(1)) if m else 0
        t = l.strip()
        if not re.match(r'^(v_|s_|ds_|global_|scratch_|buffer_|flat_)', t): continue
        op = t.split()[0]
        c[cur]['all'] += 1
        for key, hit in (('valu', op.startswith('v_')), ('v_mov', op.startswith('v_mov_b32')), ('s_nop', op == 's_nop'), ('scratch', op.startswith('scratch_')),
                         ('sgpr_spill_moves', op in ('v_readlane_b32', 'v_writelane_b32')), ('branches', op.startswith('s_cbranch')), ('dpp', 'dpp' in t)):
            if hit: c[cur][key] += 1
    print(name)
    for d in sorted(c): print("  loop depth %d: %s" % (d, dict(c[d])))
PY
make -s resource-usage EXTRA="$*" 2>&1 | grep -A9 "trace_paths_kernelILb0" | grep -E "VGPRs:|ScratchSize|Spill" | sed "s/.*remark: *//"
rm -rf $TMP
