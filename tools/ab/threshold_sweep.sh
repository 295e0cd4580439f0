cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r05_t1_threshold_sweep.txt}; : > $OUT
for S in 12 16 20 24; do for F in 4; do
  echo -n "SHADE=$S FILL=$F  C2 batch32: " >> $OUT; PT_TUNE_SHADE=$S PT_TUNE_FILL=$F PB_BATCH=32 PB_VARY=1 python3 tools/pipeline_bench.py 192 2>&1 | tail -1 >> $OUT
  echo -n "SHADE=$S FILL=$F  C4        : " >> $OUT; PT_TUNE_SHADE=$S PT_TUNE_FILL=$F PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 python3 tools/pipeline_bench.py 32 2>&1 | tail -1 >> $OUT
done; done
for F in 2 6 8; do S=16
  echo -n "SHADE=$S FILL=$F  C2 batch32: " >> $OUT; PT_TUNE_SHADE=$S PT_TUNE_FILL=$F PB_BATCH=32 PB_VARY=1 python3 tools/pipeline_bench.py 192 2>&1 | tail -1 >> $OUT
  echo -n "SHADE=$S FILL=$F  C4        : " >> $OUT; PT_TUNE_SHADE=$S PT_TUNE_FILL=$F PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 python3 tools/pipeline_bench.py 32 2>&1 | tail -1 >> $OUT
done
cat $OUT
