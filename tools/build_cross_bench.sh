#!/bin/bash
# builds tools/cross_bench (gfx950) into exp/ (git-ignored; travels to the GPU box with the snapshot)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/exp
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wall "$@" -o $R/exp/cross_bench $R/tools/cross_bench.hip
