#!/bin/bash
# A/B of two builds of libspx on the SAME box, interleaved: tools/r2/ab_libs.sh <op> libspx_a.so libspx.so [rounds]
op="$1"; a="$2"; b="$3"; rounds="${4:-3}"
for r in $(seq 1 "$rounds"); do
  for lib in "$a" "$b"; do
    printf "%-18s " "$lib"; SPX_LIB_NAME="$lib" SPX_NO_BUILD=1 python tools/bench_op.py "$op" 20 2>/dev/null | tail -1
  done
done
