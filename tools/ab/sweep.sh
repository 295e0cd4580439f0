#!/bin/bash
# One parameterised sweep instead of a script per question: time C2 (32-frame launches and one render() per frame) and C4 for every
# value of one launch heuristic, in one gpurun session (box-to-box variance is ~10 %, so only numbers of one session compare).
#   usage: tools/ab/sweep.sh KNOB v1 v2 ...        KNOB = a PT_TUNE_* name without the prefix (SHADE FILL CHUNK ROWS XCD SLOTS
#                                               GRIDDIV QUAD CULL) or BUILD for -D flags (the library is rebuilt per value)
#   e.g.   gpurun -- 'tools/ab/sweep.sh SHADE 8 16 24'      gpurun -- 'tools/ab/sweep.sh BUILD "-DPT_SHORT_STACK=8" "-DPT_MEGA_WAVES_PER_SIMD=5"'
cd $GRAFT_REPO_ROOT
KNOB=$1; shift
for V in "$@"; do
  echo "== $KNOB = $V"
  if [ "$KNOB" = BUILD ]; then
    make -s -B -j8 -C raytracer-public_amd/csrc EXTRA="$V" 2>&1 | grep -E "error" && continue
    bash tools/ab/quick_ab.sh
  else
    env PT_TUNE_$KNOB=$V bash tools/ab/quick_ab.sh
  fi
done
[ "$KNOB" = BUILD ] && make -s -B -j8 -C raytracer-public_amd/csrc 2>&1 | grep -E "error"
exit 0
