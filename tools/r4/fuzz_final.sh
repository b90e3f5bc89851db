#!/bin/bash
# Round 4, third session: the fuzz tools of rounds 1-3 that touch what this session changed (psi(y), prox + value, Binf groups and
# their deferred list) plus the general ones, on the final build.  One line per tool; each tool compares with the CPU oracle.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export SPX_NO_BUILD=1
OUT=gpurun_out/r4b/fuzz_final; mkdir -p "$OUT"; : > "$OUT/summary.txt"
for f in fuzz_obj_box fuzz_misc_scenarios fuzz_binf_many fuzz_binf_scenarios fuzz_binf_ties fuzz_binf_reversed fuzz_binf_outside_tr fuzz_ragged_many fuzz_many_instances fuzz_lattice_separable fuzz_topr_seeds fuzz_b2_scenarios; do
  start=$(date +%s)
  timeout -k 10 200 python tools/$f.py > "$OUT/$f.log" 2>&1; rc=$?
  echo "== $f: rc=$rc $(( $(date +%s) - start )) s | $(tail -n 2 "$OUT/$f.log" | tr '\n' ' ' | cut -c1-260)" | tee -a "$OUT/summary.txt"
done
