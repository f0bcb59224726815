#!/bin/bash
# builds and runs every micro-benchmark of this directory on the GPU box; each under its own timeout
set -u
cd "$(dirname "$0")"
mkdir -p /tmp/lsc_microbench
for f in *.hip; do
  b=/tmp/lsc_microbench/${f%.hip}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $b $f 2>/dev/null || { echo "$f: build failed"; continue; }
  echo "== $f"
  timeout 60 $b
done
