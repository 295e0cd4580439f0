#!/bin/bash
# bench.py batch rule for short runs
cd $GRAFT_REPO_ROOT
for KB in "20 2" "20 3" "20 5" "20 10" "50 6" "50 8" "50 12" "50 16" "50 25" "100 12" "100 16" "100 25" "100 32"; do
  set -- $KB
  echo -n "steps=$1 batch=$2: "
  PT_BENCH_BATCH=$2 timeout -k 10 120 python bench.py --steps $1 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])" || exit 1
done
