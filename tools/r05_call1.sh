#!/bin/bash
# round 5, GPU call 1: tests, entropy-decoder experiments, API trace, decode stage timeline, the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export PYTHONPATH=$R
timeout 900 python -m pytest tests -q -m gpu -rf -p no:cacheprovider > gpurun_out/r05_pytest_gpu_1.log 2>&1; tail -30 gpurun_out/r05_pytest_gpu_1.log | cut -c1-300
timeout 600 python tools/huff_exp.py > gpurun_out/r05_huff_exp.txt 2> gpurun_out/r05_huff_exp.err; cat gpurun_out/r05_huff_exp.txt | cut -c1-220
timeout 300 python tools/trace_api.py 2> gpurun_out/r05_api_trace.txt; grep -c "device" gpurun_out/r05_api_trace.txt
UHDR_HIP_CLOCK_DEBUG=1 timeout 300 python tools/decode_stages.py > gpurun_out/r05_decode_stages.txt 2>&1; grep -v "^uhdr_hip:" gpurun_out/r05_decode_stages.txt | tail -6 | cut -c1-250
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_1.json 2> gpurun_out/r05_bench_1.err; tail -c 1500 gpurun_out/r05_bench_1.json; tail -5 gpurun_out/r05_bench_1.err | cut -c1-300
