#!/bin/bash
cd $GRAFT_REPO_ROOT
export PB_BATCH=8
for K in 1 4 8 16; do echo -n "leaf=$K: "; PT_TUNE_LEAF=$K timeout -k 10 60 python tools/pipeline_bench.py 240; done
