#!/bin/bash
# builds tools/r4/c_latency.c against the in-tree libspx.so and runs it (on a GPU box): tools/r4/c_latency.sh [output file]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
ROCM="${ROCM_PATH:-/opt/rocm}"
LIB="$ROOT/shiftedproximaloperators.jl_amd/lib"
mkdir -p "$ROOT/gpurun_out"
gcc -std=c11 -O2 -Wall -D__HIP_PLATFORM_AMD__ -I"$ROOT/include" -I"$ROCM/include" "$ROOT/tools/r4/c_latency.c" \
    -L"$LIB" -lspx -Wl,-rpath,"$LIB" -L"$ROCM/lib" -lamdhip64 -Wl,-rpath,"$ROCM/lib" -lm -o "$ROOT/gpurun_out/c_latency"
"$ROOT/gpurun_out/c_latency"
