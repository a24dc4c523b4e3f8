#!/bin/bash
# round 5, GPU call 2: entropy decoder after the compaction of pass 1, spill A/B, sectioned profile, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_huffman_sync.py tests/test_gpu_jpeg_decode.py tests/test_gpu_facade.py tests/test_gpu_fuzz.py -q -rf -p no:cacheprovider > gpurun_out/r05_pytest_gpu_2.log 2>&1; tail -12 gpurun_out/r05_pytest_gpu_2.log | cut -c1-300
timeout 600 python tools/huff_exp.py > gpurun_out/r05_huff_exp2.txt 2> gpurun_out/r05_huff_exp2.err; cat gpurun_out/r05_huff_exp2.txt | cut -c1-220
UHDR_HIP_SPILL_WPE3=0 timeout 300 python tools/spill_exp.py > gpurun_out/r05_spill_wpe6.txt 2>&1; UHDR_HIP_SPILL_WPE3=1 timeout 300 python tools/spill_exp.py > gpurun_out/r05_spill_wpe3.txt 2>&1; paste -d'\n' gpurun_out/r05_spill_wpe6.txt gpurun_out/r05_spill_wpe3.txt | grep -v amdgpu | cut -c1-150
timeout 300 python tools/trace_api.py 2> gpurun_out/r05_api_trace2.txt; grep -v amdgpu gpurun_out/r05_api_trace2.txt | sed -n 20,45p
PMC=1 LIMIT=300 PROF_DIR=r05_prof bash tools/profile_all.sh 8kC 8kB 8kA 4kAhlg b32hlg tm4k gen4k api1f api1f8k api0f fdct4k idct4k huff4k > gpurun_out/r05_prof.log 2>&1; head -60 gpurun_out/r05_prof/summary.txt | cut -c1-200
