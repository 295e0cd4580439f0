cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_s1_share_rows_ab.txt; : > $OUT
sim() { python3 tools/shard_schedule_sim.py "$@" 2>&1 | grep -v "^$\|amdgpu.ids" >> $OUT; }
for R in default 20 80 320; do
  echo "== N=8, one launch of 20 share-frames, PT_TUNE_ROWS=$R" >> $OUT
  if [ $R = default ]; then unset PT_TUNE_ROWS; else export PT_TUNE_ROWS=$R; fi
  sim --gpus 8 --steps 20 --warmup 5 --schedule 20 --ranks 0,3
done
for R in default 5 20 80; do
  echo "== N=8, launches 5,5,5,5, PT_TUNE_ROWS=$R" >> $OUT
  if [ $R = default ]; then unset PT_TUNE_ROWS; else export PT_TUNE_ROWS=$R; fi
  sim --gpus 8 --steps 20 --warmup 5 --ranks 0,3
done
for R in default 20 80; do
  echo "== N=2, one launch of 20 halves, PT_TUNE_ROWS=$R" >> $OUT
  if [ $R = default ]; then unset PT_TUNE_ROWS; else export PT_TUNE_ROWS=$R; fi
  sim --gpus 2 --steps 20 --warmup 5 --schedule 20 --ranks 0
done
cat $OUT
