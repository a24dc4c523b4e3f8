#!/bin/bash
# Round-end evidence in one gpurun call: smoke, the driver's bench command, the gloo two-rank dry run of the N > 1 bench
# path, HEAD-state rocprofv3 summaries (their databases stay in /tmp on the box).  Everything lands under gpurun_out/;
# copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export PYTHONPATH=$R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1; tail -1 gpurun_out/final_smoke.log
if [ "$1" != "nopytest" ]; then
  timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04_pytest_gpu.log 2>&1; tail -1 gpurun_out/r04_pytest_gpu.log
  timeout 200 python tests/fuzz_parity.py --seconds 120 --seed 43 > gpurun_out/r04_fuzz_parity.log 2>&1; tail -1 gpurun_out/r04_fuzz_parity.log | cut -c1-160
fi
timeout 300 python tools/trace_api.py 2> gpurun_out/r04_api_trace.txt; grep -c "device" gpurun_out/r04_api_trace.txt
timeout 300 python tools/decode_stages.py > gpurun_out/r04_decode_stages.txt 2>&1; tail -5 gpurun_out/r04_decode_stages.txt | cut -c1-220
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_head.json 2> gpurun_out/r04_bench_head.err; tail -c 300 gpurun_out/r04_bench_head.json
UHDR_BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --no-extra --no-cpu > gpurun_out/r04_bench_gloo_2ranks.json 2> gpurun_out/r04_bench_gloo_2ranks.err; tail -c 200 gpurun_out/r04_bench_gloo_2ranks.json
[ "$1" = "noprof" ] && exit 0
PROF_DIR=prof_all4 bash tools/profile_all.sh > gpurun_out/prof_all4.log 2>&1; tail -2 gpurun_out/prof_all4.log | cut -c1-160
cd $R
[ "$1" = "nobenchprof" ] && exit 0
bash tools/profile_bench.sh > gpurun_out/prof_bench4.log 2>&1; tail -4 gpurun_out/prof_bench4.log | cut -c1-200
du -sh gpurun_out
