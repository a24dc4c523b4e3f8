#!/bin/bash
# builds kbench variants (experiment macros) -- run on the build machine; binaries go to tools/kb_*
cd "$(dirname "$0")"
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../libultrahdr_amd/csrc -I../include -Wno-unused-variable"
for v in base "$@"; do
  D=""; [ "$v" != base ] && D="-DUHDR_EXP_$v"
  /opt/rocm/bin/hipcc $FL $D -o kb_$v kbench.cpp 2>&1 | grep -E "error" &
done
wait
ls -la kb_*
