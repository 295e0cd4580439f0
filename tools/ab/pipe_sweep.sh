#!/bin/bash
# one render() per frame, no host waits (the reference's call shape): frame slots x grid size
cd $GRAFT_REPO_ROOT
for S in ${SLOTS:-2 3 4 6 8}; do for G in ${GRIDS:-1 2 3}; do
  echo -n "slots=$S griddiv=$G: "; PT_TUNE_SLOTS=$S PT_TUNE_GRIDDIV=$G PB_BATCH=${PB_BATCH:-1} PB_VARY=1 python3 tools/pipeline_bench.py ${FRAMES:-60} 2>&1 | tail -1
done; done
