#!/bin/bash
# PT_MODE_REFERENCE (the reference's own frame) with one render() per launch over frame slots x grid divisors x queue rows
cd $GRAFT_REPO_ROOT
for S in ${SLOTS:-3 6 8}; do for G in ${GRIDS:-1 2 4 8}; do for R in ${ROWSS:-0}; do
  echo -n "slots=$S griddiv=$G rows=$R: "
  if [ "$R" = 0 ]; then PT_TUNE_SLOTS=$S PT_TUNE_GRIDDIV=$G python3 tools/reference_mode_fps.py 2>&1 | head -1
  else PT_TUNE_ROWS=$R PT_TUNE_SLOTS=$S PT_TUNE_GRIDDIV=$G python3 tools/reference_mode_fps.py 2>&1 | head -1; fi
done; done; done
