cd $GRAFT_REPO_ROOT
OUT=gpurun_out/m3_pieces.txt; : > $OUT
sim() { python3 tools/shard_schedule_sim.py "$@" 2>&1 | grep -v "^$" | grep -v amdgpu >> $OUT; }
for N in 2 4 8; do for S in 5,5,5,5 7,7,6 10,10 20; do sim --gpus $N --steps 20 --warmup 5 --schedule $S --ranks 0,1; done; done
grep -E "launches|slowest" $OUT
