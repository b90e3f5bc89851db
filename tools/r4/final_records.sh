#!/bin/bash
# The round's records, on the FINAL code, in one gpurun call: headline summary (stats + both PMC passes), every operator's kernel
# stats and traffic, the team form's, the bench line.  -> gpurun_out/r4/final/
set -uo pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F=gpurun_out/r4/final; mkdir -p "$F"
python3 -c "import __graft_entry__ as g; g.build()" > "$F/build.log" 2>&1
bash tools/profile.sh > "$F/l1box_summary.txt" 2>&1; echo "headline profile done"
bash tools/profile_ops.sh > "$F/all_ops_kernel_stats.txt" 2>&1; echo "all-ops stats done"
bash tools/profile_ops_traffic.sh > "$F/traffic_all_ops.txt" 2>&1; echo "all-ops traffic done"
python3 bench.py > "$F/bench.json" 2> "$F/bench.err"; echo "bench done"
python3 tools/r3/check_bench_kernels.py "$F/bench.json" "$F/all_ops_kernel_stats.txt" > "$F/check_bench_kernels.txt" 2>&1; cat "$F/check_bench_kernels.txt"
tail -3 "$F/l1box_summary.txt"
