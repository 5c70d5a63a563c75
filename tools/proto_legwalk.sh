#!/bin/bash
# Builds and runs the sub-configuration-parallelism prototype (tools/proto_legwalk.hip) on the GPU box; output under gpurun_out/.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/exp gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize -mllvm -disable-machine-licm \
   -o build/exp/proto_legwalk tools/proto_legwalk.hip
for groups in 64 256; do
   timeout -k 10 120 ./build/exp/proto_legwalk $groups
done | tee gpurun_out/r05_proto_legwalk.txt
