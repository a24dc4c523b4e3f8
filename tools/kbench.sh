#!/bin/bash
# builds kbench variants -- run on the build machine; binaries go to tools/kb_<name>
#   kbench.sh NAME[:-Dflag[,-Dflag...]] ...     (a bare NAME means -DUHDR_EXP_NAME)
cd "$(dirname "$0")"
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../libultrahdr_amd/csrc -I../include -Wno-unused-variable"
for v in base "$@"; do
  name=${v%%:*}; D=""
  if [ "$v" != base ]; then
    if [[ "$v" == *:* ]]; then D=$(echo "${v#*:}" | tr ',' ' '); else D="-DUHDR_EXP_$v"; fi
  fi
  /opt/rocm/bin/hipcc $FL $D -o kb_$name kbench.cpp 2>&1 | grep -E "error" &
done
wait
ls kb_*
