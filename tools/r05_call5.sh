#!/bin/bash
# round 5, call 5: the two-scan entry points -- parity tests, then the API-1 round trip with and without them
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_huffman_sync.py tests/test_gpu_jpeg_decode.py tests/test_gpu_facade.py tests/test_zz_device_chains.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_pytest_gpu_5.log
cat gpurun_out/r05_pytest_gpu_5.log
timeout 600 python - <<'PY' > gpurun_out/r05_roundtrip_5.json 2> gpurun_out/r05_roundtrip_5.err
import json, torch
import bench
from libultrahdr_amd.ultrahdr import Context, UltraHdr
torch.cuda.set_device(0)
ctx = Context(0)
u = UltraHdr(ctx=ctx)

r = bench.api1_roundtrip_section(ctx, u, "cuda:0")
print(json.dumps({k: v for k, v in r.items() if not isinstance(v, dict)}, indent=1))
r = bench.api1_roundtrip_section(ctx, u, "cuda:0")
print(json.dumps({k: v for k, v in r.items() if not isinstance(v, dict)}, indent=1))
PY
cat gpurun_out/r05_roundtrip_5.json; tail -5 gpurun_out/r05_roundtrip_5.err
