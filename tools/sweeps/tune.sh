#!/bin/bash
# sweep runtime knobs of the megakernel: tools/tune.sh  (prints kernel ms per setting)
cd $GRAFT_REPO_ROOT
for P in 0; do for F in 32; do
  echo -n "passes=$P flush=$F : "; PT_TUNE_PASSES=$P PT_TUNE_FLUSH=$F timeout -k 10 60 python tools/profile_frame.py 4 | tail -1
done; done
for SH in 4 16; do for FI in 4 16; do
  echo -n "shade=$SH fill=$FI : "; PT_TUNE_SHADE=$SH PT_TUNE_FILL=$FI timeout -k 10 60 python tools/profile_frame.py 4 | tail -1
done; done
for C in 256; do echo -n "chunk=$C : "; PT_TUNE_CHUNK=$C timeout -k 10 60 python tools/profile_frame.py 4 | tail -1; done
