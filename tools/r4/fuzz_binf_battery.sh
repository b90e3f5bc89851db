#!/bin/bash
# the Binf fuzz tools of rounds 1-2 on the current build (after the round-4 changes of binf_root): one file per tool under gpurun_out/r4b/fuzz
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4b/fuzz; mkdir -p "$OUT"
export SPX_NO_BUILD=1
for t in fuzz_binf_many fuzz_binf_scenarios fuzz_binf_ties fuzz_binf_reversed fuzz_binf_outside_tr fuzz_binf_mid; do
  echo "== $t"; timeout -k 10 400 python tools/$t.py > "$OUT/$t.txt" 2>&1; echo "rc $?"; tail -2 "$OUT/$t.txt"
done
echo "== fuzz_r2_binf_random 5000 400"; timeout -k 10 400 python tools/fuzz_r2_binf_random.py 5000 400 > "$OUT/fuzz_r2_binf_random.txt" 2>&1; echo "rc $?"; tail -2 "$OUT/fuzz_r2_binf_random.txt"
echo "== fuzz_r2_binf_lattice 5000 300"; timeout -k 10 400 python tools/fuzz_r2_binf_lattice.py 5000 300 > "$OUT/fuzz_r2_binf_lattice.txt" 2>&1; echo "rc $?"; tail -2 "$OUT/fuzz_r2_binf_lattice.txt"
