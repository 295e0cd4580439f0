#!/bin/bash
# one line per workload for the current build and PT_TUNE_* environment: C2 (32-frame launches), C2 one render() per frame, C4
cd $GRAFT_REPO_ROOT
PB_BATCH=32 PB_VARY=1 python3 tools/pipeline_bench.py 128 2>&1 | tail -1
PB_BATCH=1 PB_VARY=1 python3 tools/pipeline_bench.py 40 2>&1 | tail -1
PB_SOLO=1 PB_BATCH=1 PB_VARY=1 python3 tools/pipeline_bench.py 40 2>&1 | tail -1 | sed 's/^/solo /'
PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 python3 tools/pipeline_bench.py 24 2>&1 | tail -1
