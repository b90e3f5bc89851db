#!/bin/bash
# rocprofv3 recipe for the headline kernel (run on the GPU box through gpurun).  Kernel trace/stats and the
# PMC passes are separate runs; FETCH_SIZE and WRITE_SIZE cannot share a pass (MI355X_MICROARCH.md, PMC slots).
set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof; rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="bench.py --steps 20 --warmup 3 --no-cpu --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/stats.log" 2>&1 || { echo "stats run failed"; tail -5 "$OUT/stats.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $ARGS > "$OUT/fetch.log" 2>&1 || { echo "fetch run failed"; tail -5 "$OUT/fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $ARGS > "$OUT/write.log" 2>&1 || { echo "write run failed"; tail -5 "$OUT/write.log"; exit 1; }
find "$OUT" -name '*.csv' | head -20
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1; cat "$OUT/summary.txt"
