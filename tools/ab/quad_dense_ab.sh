#!/bin/bash
# Round 4, VERDICT item 1: is one ray per QUAD of lanes (one child box and one 16-byte request per lane) also the better shape for the DENSE part
# of a launch?  Same-session A/B of two builds -- the default (one ray per lane while there is work to start, quads for the drain) and
# -DPT_QUAD=2 (quads from the first ray on) -- with timings and the counters the question is about: vector-L1 accesses
# (TCP_TOTAL_CACHE_ACCESSES), vector / scalar instructions, lane utilisation, per frame of 32-frame C2 launches (rocprofv3 --pmc passes,
# never combined with tracing).   usage (GPU box): tools/ab/quad_dense_ab.sh <out dir under gpurun_out>
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT; : > $OUT/summary.txt
export TMPDIR=/tmp
for V in "" "-DPT_QUAD=2"; do
  TAG=$( [ -z "$V" ] && echo default || echo quad_always )
  echo "== build: ${V:-<default>}" >> $OUT/summary.txt
  make -s -B -j8 -C raytracer-public_amd/csrc EXTRA="$V" 2>&1 | grep -E "error" >> $OUT/summary.txt && continue
  # the build under test is bit-exact before it is timed (a wrong kernel's time means nothing)
  timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -q -x -k "path_mode_bit_exact or sponza_class or stack_overflow" 2>&1 | tail -1 >> $OUT/summary.txt
  { echo -n "batch32   "; PB_BATCH=32 PB_VARY=1 timeout -k 10 120 python3 tools/pipeline_bench.py 128 2>&1 | tail -1
    echo -n "solo1     "; PB_SOLO=1 PB_BATCH=1 PB_VARY=1 timeout -k 10 120 python3 tools/pipeline_bench.py 40 2>&1 | tail -1
    echo -n "sponza    "; PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 timeout -k 10 120 python3 tools/pipeline_bench.py 24 2>&1 | tail -1
  } >> $OUT/summary.txt
  mkdir -p $OUT/$TAG; i=0
  for CNT in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum"; do
    i=$((i+1))
    ( cd /tmp && PB_BATCH=32 PB_VARY=1 timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d $OUT/$TAG/pass$i -- python3 $GRAFT_REPO_ROOT/tools/pipeline_bench.py 32 > $OUT/$TAG/pass$i.log 2>&1 ) || echo "pass $i failed" >> $OUT/summary.txt
  done
  python3 - "$OUT/$TAG" >> $OUT/summary.txt <<'PY'
import csv, glob, collections, sys
res = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "trace_paths" not in row.get("Kernel_Name", ""): continue
        res.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for k, v in res.items():
    print("  %-32s %8.1f M per frame (%d launches of 32 frames)" % (k, sum(v) / len(v) / 32.0 / 1e6, len(v)))
if "SQ_THREAD_CYCLES_VALU" in res and "SQ_ACTIVE_INST_VALU" in res:
    print("  lane utilisation %.3f" % (sum(res["SQ_THREAD_CYCLES_VALU"]) / (64.0 * sum(res["SQ_ACTIVE_INST_VALU"]))))
PY
done
make -s -B -j8 -C raytracer-public_amd/csrc 2>&1 | grep -E "error"
cat $OUT/summary.txt
