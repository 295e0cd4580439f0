#!/bin/bash
# The counter side of bench.py's roofline: rocprofv3 --pmc passes (one counter group per run, never combined with tracing) and one
# --kernel-trace --stats run over THE COMMAND THE DRIVER RUNS (`python3 bench.py --steps 20 --warmup 5`), reduced to per-frame totals of
# the timed trace_paths_kernel launches -> profiles/r06_pmc_bench.json (+ kernel stats CSV; the caller copies them there).  bench.py writes the order of its launches
# (warm-up / timed / reference call shape) to a side file, so the timed dispatches are picked by position, not guessed.
# usage (on the GPU box): tools/pmc_bench.sh <outdir-under-gpurun_out> [bench.py arguments, default: --steps 20 --warmup 5]
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$1; shift
ARGS=${*:---steps 20 --warmup 5}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PT_BENCH_LAUNCH_LOG=$OUT/launch_log.json
i=0
# (PMC_GROUPS="counters of pass 1;counters of pass 2;..." replaces the groups below: e.g. the wavefront-time breakdown of profiles/r05_pmc_issue.json)
GROUPS_DEFAULT="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY;FETCH_SIZE;WRITE_SIZE TCC_HIT_sum TCC_MISS_sum;GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS;TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
IFS=';' read -r -a GROUPS_ARR <<< "${PMC_GROUPS:-$GROUPS_DEFAULT}"
for CNT in "${GROUPS_ARR[@]}"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline --no-configs --sustained-seconds 0 > $OUT/pass$i.log 2>&1 || echo "pass $i ($CNT) failed" >> $OUT/errors.txt
  cp $OUT/launch_log.json $OUT/launch_log_pass$i.json 2>/dev/null
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline --no-configs --sustained-seconds 0 > $OUT/trace.log 2>&1 || echo "kernel-trace run failed" >> $OUT/errors.txt
cp $OUT/launch_log.json $OUT/launch_log_trace.json 2>/dev/null
python3 $ROOT/tools/pmc_bench_reduce.py $OUT "python3 bench.py $ARGS"
