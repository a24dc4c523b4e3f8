#!/bin/bash
# round 5, GPU call 3: whole GPU suite (sparse Huffman walk, ADVICE fixes), fuzz, the bench line with all new sections
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export PYTHONPATH=$R
timeout 900 python -m pytest tests -q -m gpu -rf -p no:cacheprovider > gpurun_out/r05_pytest_gpu_3.log 2>&1; tail -15 gpurun_out/r05_pytest_gpu_3.log | cut -c1-300
timeout 200 python tests/fuzz_parity.py --seconds 90 --seed 51 > gpurun_out/r05_fuzz_parity_3.log 2>&1; tail -2 gpurun_out/r05_fuzz_parity_3.log | cut -c1-200
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_3.json 2> gpurun_out/r05_bench_3.err; tail -c 600 gpurun_out/r05_bench_3.json; grep -v "uhdr_hip_seam\|amdgpu" gpurun_out/r05_bench_3.err | tail -5 | cut -c1-300
