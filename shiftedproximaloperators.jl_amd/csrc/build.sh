#!/bin/bash
# Builds libspx.so (HIP, gfx950 only) next to the package: shiftedproximaloperators.jl_amd/lib/libspx.so
# -ffp-contract=off: the reference (Julia) never fuses a*b+c; several kernels decide branches on such sums.
# One object per source file, compiled in parallel (no cross-file device calls, so no -fgpu-rdc), then one link.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../lib"
NAME="${SPX_LIB_NAME:-libspx.so}"
OBJ="$OUT/obj_${NAME%.so}"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden
       -Wall -Wno-unused-variable -Wno-unused-but-set-variable -I"$ROOT/include" -I"$HERE" "$@")
SRCS=(spx_ctx spx_separable spx_separable_f32 spx_select spx_group spx_group_team spx_group_f32 spx_objective spx_b2 spx_host)
pids=()
objs=()
for s in "${SRCS[@]}"; do
  "$HIPCC" "${FLAGS[@]}" -c "$HERE/$s.hip" -o "$OBJ/$s.o" &
  pids+=($!)
  objs+=("$OBJ/$s.o")
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
[ "$fail" = 0 ] || { echo "compile failed" >&2; exit 1; }
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden "${objs[@]}" -o "$OUT/$NAME.tmp"
mv -f "$OUT/$NAME.tmp" "$OUT/$NAME"
echo "built $OUT/$NAME"
