#!/bin/bash
# HBM traffic of a batched (32-frame) trace_paths_kernel launch: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (never combined with tracing).  usage: tools/ab/pmc_batched.sh <outdir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PB_BATCH=32 PB_VARY=1      # 32 different frames (frame indices) per launch, as bench.py submits them
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass$i -- python $GRAFT_REPO_ROOT/tools/pipeline_bench.py 32 > $OUT/pass$i.log 2>&1 || echo "pass $i failed" >> $OUT/errors.txt
done
python3 - <<PY
import csv, glob, collections, json
res = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "trace_paths" not in row.get("Kernel_Name", ""): continue
        res.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summary = {k: {"launches": len(v), "mean_per_launch": sum(v) / len(v), "mean_per_frame": sum(v) / len(v) / 32.0} for k, v in res.items()}
json.dump(summary, open("$OUT/pmc_batched_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
