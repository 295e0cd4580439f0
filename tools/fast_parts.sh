#!/bin/bash
# which part of the FAST variant changes the image / buys the time: all of it, or only the fused slab + v_rcp (exact child order kept)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; : > $OUT
for V in "" "-DPT_FAST_ORDER=0"; do
  echo "== build: ${V:-<FAST = fused slab + v_rcp + plain child order>}" >> $OUT
  make -s -B -j8 -C raytracer-public_amd/csrc EXTRA="$V" 2>&1 | grep -E "error" >> $OUT
  timeout -k 10 200 python3 tools/fast_check.py 2>&1 | grep -v amdgpu.ids >> $OUT
  for F in 0 1; do echo -n "FAST=$F batch32 " >> $OUT; PT_TUNE_FAST=$F PB_BATCH=32 PB_VARY=1 python3 tools/pipeline_bench.py 128 2>&1 | tail -1 >> $OUT
                   echo -n "FAST=$F sponza  " >> $OUT; PT_TUNE_FAST=$F PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 python3 tools/pipeline_bench.py 24 2>&1 | tail -1 >> $OUT; done
done
make -s -B -j8 -C raytracer-public_amd/csrc 2>&1 | grep -E "error"
cat $OUT
