#!/bin/bash
# Which unit limits trace_paths_kernel?  rocprofv3 --pmc passes (one counter group per run, never combined with tracing) over
# a 32-frame launch of tools/pipeline_bench.py.  usage: tools/ab/pmc_discover.sh <outdir-under-gpurun_out> [scene]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PB_BATCH=${PB_BATCH:-32} PB_VARY=1
[ "${2:-dragon}" = "sponza" ] && export PF_SCENE=sponza PB_BATCH=4
rocprofv3 -L > $OUT/counters.txt 2>&1
i=0
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TD_TD_BUSY_sum TCP_TOTAL_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/tools/pipeline_bench.py $((PB_BATCH * 3)) > $OUT/pass$i.log 2>&1 || echo "pass $i ($CNT) failed" >> $OUT/errors.txt
done
python3 - <<PY
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench                                   # only for source_tag(): the hash of the kernel sources these counters belong to
B = float(os.environ["PB_BATCH"])
res = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name", "")
        if "trace_paths" not in kn and "trace2" not in kn: continue
        res.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
# only full launches (the warm-up runs full batches too); per-frame means
summary = {k: {"launches": len(v), "mean_per_frame": sum(v) / len(v) / B} for k, v in res.items()}
c = {k: v["mean_per_frame"] for k, v in summary.items()}
derived = {}
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c: derived["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
if "GRBM_GUI_ACTIVE" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0           # the counter sums over the 8 XCDs
    derived["gpu_cycles_per_frame"] = cyc
    if "SQ_INSTS_VALU" in c: derived["valu_issue_fraction"] = c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c: derived["l1_requests_per_cycle_and_cu"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / cyc
summary = {"source_tag": bench.source_tag(), "scene": "${2:-dragon}", "frames_per_launch": B, "derived": derived, "counters": summary}
json.dump(summary, open("$OUT/pmc_discover_summary.json", "w"), indent=1)
print(json.dumps(derived, indent=1))
summary = summary["counters"]
for k, v in summary.items(): print("%-40s %16.1f  (%d launches)" % (k, v["mean_per_frame"], v["launches"]))
PY
