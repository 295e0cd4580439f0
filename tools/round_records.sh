#!/bin/bash
# The end-of-round records of one session (GPU box): the driver's command, the default run, every BASELINE configuration, wave timelines, the reference's own workload,
# the launch stress, scene build times -> gpurun_out/r06_*; the caller copies what it keeps into profiles/.
cd $GRAFT_REPO_ROOT
# the counter passes FIRST: bench.py prints roofline fractions only from a counter file whose tag matches the kernel sources it runs
tools/pmc_bench.sh r06_pmc > gpurun_out/r06_pmc.log 2>&1; cp gpurun_out/r06_pmc/pmc_bench.json profiles/r06_pmc_bench.json && echo "pmc ok"
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_n1_steps20.json 2> gpurun_out/r06_bench_n1_steps20.err; echo "bench20 rc=$?"
timeout -k 10 300 python bench.py > gpurun_out/r06_bench_n1_default.json 2> gpurun_out/r06_bench_n1_default.err; echo "bench256 rc=$?"
timeout -k 10 200 python3 tools/config_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_configs.txt; cat gpurun_out/r06_configs.txt
timeout -k 10 120 python3 tools/wave_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_wave_timeline_lone.txt; echo "wt rc=$?"
WT_BATCH=20 timeout -k 10 120 python3 tools/wave_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_wave_timeline_20frames.txt; echo "wt20 rc=$?"
timeout -k 10 200 python3 tools/reference_mode_fps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_reference_mode_fps.txt; cat gpurun_out/r06_reference_mode_fps.txt
timeout -k 10 200 python3 tools/launch_stress.py 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r06_launch_stress.txt; cat gpurun_out/r06_launch_stress.txt
timeout -k 10 100 python3 tools/build_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_build_time.txt; cat gpurun_out/r06_build_time.txt
