#!/bin/bash
# One-GPU replay of every rank's submission sequence for the driver's sharded commands (tools/shard_schedule_sim.py), for the
# schedule bench.py uses and for candidate schedules, in ONE session (box-to-box variance is ~10 %).
#   usage (GPU box): tools/ab/shard_schedule_sweep.sh <out file under gpurun_out>
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $(dirname $OUT); : > $OUT
sim() { python3 tools/shard_schedule_sim.py "$@" 2>&1 | grep -v "^$" >> $OUT; }
echo "== steps 20 warmup 5 (the driver's command), bench.py's schedule" >> $OUT
for N in 2 4 8; do sim --gpus $N --steps 20 --warmup 5; done
echo "== steps 20 warmup 5, N = 8: candidate schedules (rank 0 and 3 only)" >> $OUT
for S in 20 10,10 10,8,2 7,7,6 8,6,4,2 5,5,5,5 4,4,4,4,4 3,3,3,3,2,2,2,2; do sim --gpus 8 --steps 20 --warmup 5 --schedule $S --ranks 0,3; done
echo "== steps 20 warmup 5, N = 2 and 4: candidate schedules (rank 0 only)" >> $OUT
for S in 20 10,10 5,5,5,5; do sim --gpus 2 --steps 20 --warmup 5 --schedule $S --ranks 0; sim --gpus 4 --steps 20 --warmup 5 --schedule $S --ranks 0; done
echo "== steps 256 warmup 32 (bench.py's defaults), bench.py's schedule" >> $OUT
for N in 2 4 8; do sim --gpus $N --steps 256 --warmup 32 --reps 2; done
echo "== steps 256 warmup 32, N = 8: candidates (rank 0)" >> $OUT
for S in 256 128,128 128,64,32,16,8,8 32,32,32,32,32,32,32,32; do sim --gpus 8 --steps 256 --warmup 32 --reps 2 --schedule $S --ranks 0; done
cat $OUT
