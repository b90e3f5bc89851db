#!/bin/bash
# Builds libspx.so (HIP, gfx950 only) next to the package: shiftedproximaloperators.jl_amd/lib/libspx.so
# -ffp-contract=off: the reference (Julia) never fuses a*b+c; several kernels decide branches on such sums.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRCS=("$HERE"/spx_ctx.hip "$HERE"/spx_separable.hip "$HERE"/spx_select.hip "$HERE"/spx_group.hip "$HERE"/spx_objective.hip "$HERE"/spx_b2.hip "$HERE"/spx_host.hip)
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fvisibility=hidden \
  -Wall -Wno-unused-variable -Wno-unused-but-set-variable \
  -I"$ROOT/include" -I"$HERE" "${SRCS[@]}" -o "$OUT/${SPX_LIB_NAME:-libspx.so}" "$@"
echo "built $OUT/${SPX_LIB_NAME:-libspx.so}"
