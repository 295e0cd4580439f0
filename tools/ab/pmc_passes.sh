#!/bin/bash
# rocprofv3 PMC passes over tools/profile_frame.py (each pass its own run; --pmc is never combined
# with tracing).  usage: tools/ab/pmc_passes.sh <outdir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass$i -- python $GRAFT_REPO_ROOT/tools/profile_frame.py 2 > $OUT/pass$i.log 2>&1 || echo "pass $i failed" >> $OUT/errors.txt
done
python3 - <<PY
import csv, glob, collections, json
res = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "render_rays" not in k and "trace_paths" not in k: continue
        res.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summary = {k: sum(v) / len(v) for k, v in res.items()}
json.dump(summary, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
