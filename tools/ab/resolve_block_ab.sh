#!/bin/bash
# Same-session A/B of the resolve pass's workgroup size (256 / 128 / 64 threads): does a smaller group find a wave slot sooner next to the persistent launches?
#   usage (GPU box): bash tools/ab/resolve_block_ab.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_v3_resolve_block.txt; : > $OUT
for V in "" "-DPT_RESOLVE_BLOCK=64" "-DPT_RESOLVE_BLOCK=128" ""; do
  echo "== variant: ${V:-<default 256>}" >> $OUT
  make -s -B -j8 -C raytracer-public_amd/csrc EXTRA="$V" 2>&1 | grep -E "error" >> $OUT && continue
  timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -q -x -k "path_mode_bit_exact or progressive or batched" 2>&1 | tail -1 >> $OUT
  { echo -n "batch32   "; PB_BATCH=32 PB_VARY=1 timeout -k 10 120 python3 tools/pipeline_bench.py 128 2>&1 | tail -1
    echo -n "pipe1     "; PB_BATCH=1 PB_VARY=1 timeout -k 10 120 python3 tools/pipeline_bench.py 80 2>&1 | tail -1
    echo -n "solo1     "; PB_SOLO=1 PB_BATCH=1 PB_VARY=1 timeout -k 10 120 python3 tools/pipeline_bench.py 40 2>&1 | tail -1
    echo -n "refmode1  "; PT_TUNE_SLOTS= timeout -k 10 120 python3 tools/reference_mode_fps.py 2>&1 | head -1
  } >> $OUT
  ( cd /tmp && export TMPDIR=/tmp && PB_BATCH=1 PB_VARY=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_v3_trace -- python3 $GRAFT_REPO_ROOT/tools/pipeline_bench.py 80 > /dev/null 2>&1 )
  f=$(ls -t gpurun_out/r04_v3_trace/*/*kernel_stats.csv | head -1); grep -E "resolve_kernel|trace_paths" $f | cut -d, -f1-7 >> $OUT; rm -rf gpurun_out/r04_v3_trace
done
make -s -B -j8 -C raytracer-public_amd/csrc 2>&1 | grep error
cat $OUT
