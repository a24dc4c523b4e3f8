#!/bin/bash
# Round-6 evidence in one gpurun call (everything lands under gpurun_out/; what is to be judged is copied into profiles/ afterwards):
# smoke, the whole GPU suite, a fuzz sweep, the driver's bench command (+ its full record), the gloo two-rank dry run of the N > 1 bench
# path, per-kernel tables and one timeline of the headline round trip, 20 back-to-back facade decodes, and -- on the SAME box -- the
# rocprofv3 passes: the bench's roofline kernel with its byte counters (tools/profile_bench.sh) and the sectioned per-(case, kernel) summary
# (tools/profile_all.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
RN=${ROUND:-r06}
cd $R
mkdir -p gpurun_out
export PYTHONPATH=$R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${RN}_smoke.log 2>&1; tail -1 gpurun_out/${RN}_smoke.log
if [ "$1" != "nopytest" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -rf -p no:cacheprovider > gpurun_out/${RN}_pytest_gpu.log 2>&1; tail -1 gpurun_out/${RN}_pytest_gpu.log
  timeout 200 python tests/fuzz_parity.py --seconds 120 --seed 61 > gpurun_out/${RN}_fuzz_parity.log 2>&1; tail -1 gpurun_out/${RN}_fuzz_parity.log | cut -c1-160
fi
(time python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/${RN}_bench_head.json 2> gpurun_out/${RN}_bench_head.err; tail -c 400 gpurun_out/${RN}_bench_head.json; tail -4 gpurun_out/${RN}_bench_head.err
cp bench_detail.json gpurun_out/${RN}_bench_detail.json
UHDR_BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --no-extra --no-cpu > gpurun_out/${RN}_bench_gloo_2ranks.json 2> gpurun_out/${RN}_bench_gloo_2ranks.err; tail -c 200 gpurun_out/${RN}_bench_gloo_2ranks.json
for m in "4k two" "4k seq" "8k two"; do
  set -- $m
  bash tools/profile_roundtrip.sh 12 $1 $2 > gpurun_out/${RN}_roundtrip_kernels_$1_$2.txt 2>&1; head -1 gpurun_out/${RN}_roundtrip_kernels_$1_$2.txt
  python tools/timeline.py gpurun_out/prof_rt_$1$2 5 > gpurun_out/${RN}_roundtrip_timeline_$1_$2.txt 2>&1; tail -1 gpurun_out/${RN}_roundtrip_timeline_$1_$2.txt
  rm -rf gpurun_out/prof_rt_$1$2
done
timeout 300 python tools/r06_decode_20.py 2>&1 | grep -v amdgpu > gpurun_out/${RN}_decode_20_calls.txt; tail -3 gpurun_out/${RN}_decode_20_calls.txt | cut -c1-200
timeout 120 python tools/r06_huff_debug.py 2>&1 | grep -v amdgpu | cut -c1-500 > gpurun_out/${RN}_huff_levels.txt
[ "$1" = "noprof" ] && exit 0
bash tools/profile_bench.sh > gpurun_out/${RN}_prof_bench.log 2>&1; tail -4 gpurun_out/${RN}_prof_bench.log | cut -c1-200
cd $R
LIMIT=300 PROF_DIR=${RN}_prof bash tools/profile_all.sh 8kC 8kB 8kA 4kAhlg 4kApq b32hlg tm4k gen4k gen4k1 tm8k api0f api1f api1f8k fdct4k idct4k cvt4k huff4k > gpurun_out/${RN}_prof_all.log 2>&1; tail -2 gpurun_out/${RN}_prof_all.log | cut -c1-160
cd $R
du -sh gpurun_out
