#!/bin/bash
# Confirms that the driver-run stress tests FAIL on planted faults.
#   tools/planted_faults.sh build [ids]   (here, no GPU): patched copies of csrc/ -> lib/libspx_fault<k>.so (all, or the ids listed)
#   tools/planted_faults.sh run [ids] (on the GPU box): the named tests against each faulty library; every one must fail
# Nothing in the product tree is modified: the patches are applied to a scratch copy under /tmp.
set -uo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CSRC="$ROOT/shiftedproximaloperators.jl_amd/csrc"
# id | file | sed expression | tests that must catch it
FAULTS=(
 "1|spx_group_common.hpp|0,/bool decided = g0 < -1e-9 \* sl;/s//bool decided = g0 < 1e300;/|tests/test_gpu_stress.py::test_binf_reversed_bracket_entries_outside_trust_region"
 "2|spx_group_common.hpp|s/decided = (r2n == r2);/decided = true;/g|tests/test_gpu_stress.py::test_binf_reversed_bracket_entries_outside_trust_region"
 "3|spx_group_common.hpp|s/if (reversed \&\& mX < delta \* (1.0 - 1e-9)) return BINF_ZERO;/if (reversed) return BINF_ZERO;/|tests/test_gpu_stress.py::test_binf_reversed_bracket_entries_outside_trust_region tests/test_gpu_stress.py::test_binf_reversed_bracket_regimes"
 "5|spx_select.hip|s/} else if (!catch_all) {  /} else if (true) {  /|tests/test_gpu_stress.py::test_topr_folded_first_digit"
 "6|spx_b2.hip|s/clear_rows\[pass_i \* kB2Cols \* kB2Words + rem\] = 0ull;/(void)rem;/|tests/test_gpu_stress.py::test_b2_alternating_sizes_share_the_exchange_words tests/test_gpu_stress.py::test_b2_one_launch_forms_at_their_boundaries"
 "7|spx_select.hip|s/rc = spx_zero_async(ctx, &ss->chist\[0\]\[0\]\[0\], sizeof(ss->chist\[0\]));/rc = 0;/|tests/test_gpu_graph.py::test_iteration_in_a_graph_replays_on_new_data"
 "8|spx_select.hip|s/      if (run_count\[k\]) atomicAdd(&ws->hist\[run_digit\[k\]\], (unsigned long long)run_count\[k\]);/      (void)run_count[k];/|tests/test_gpu_parity.py::test_indball_l0_front_sample_sizes tests/test_gpu_fullsize.py"
 "9|spx_select.hip|s/if (match \&\& ex + 1ull == rho) tl\[19\]/if (match \&\& ex == rho) tl[19]/|tests/test_gpu_stress.py::test_topr_tie_mode_against_exact_select tests/test_gpu_fullsize.py::test_indball_fast_path_and_fallback"
 "10|spx_b2.hip|s/for (unsigned int e = (unsigned int)(t \& 63); e < ncand_mine; e += 64) {/for (unsigned int e = (unsigned int)(t \& 63); e + 64 < ncand_mine; e += 64) {/|tests/test_gpu_stress.py::test_b2_streaming_form_scenarios"
 "11|spx_select.hip|s/const int64_t lo = (ca < cs ? ca : cs) + 1, hi = ca < cs ? cs : ca;/const int64_t lo = (ca < cs ? ca : cs) + 3, hi = ca < cs ? cs : ca;/|tests/test_gpu_stress.py::test_topr_tie_mode_against_exact_select"
 "12|spx_select.hip|s/(spec_hi == 2 \&\& i <= spec_cut);/(spec_hi == 2 \&\& i < spec_cut);/|tests/test_gpu_stress.py::test_topr_tie_mode_against_exact_select"
 "13|spx_select.hip|s/if (tail \&\& pr == nw) {  \/\/ the elements behind the last whole vector/if (false) {  \/\/ the elements behind the last whole vector/|tests/test_gpu_parity.py::test_indball_l0_at_the_fast_path_threshold"
 "14|spx_select.hip|s/if (wr \&\& !in \&\& keep != spec) y\[i\]/if (wr \&\& !in \&\& keep \&\& !spec) y[i]/|tests/test_gpu_parity.py::test_indball_l0_at_the_fast_path_threshold tests/test_gpu_parity.py::test_indball_l0_ranks_and_scales"
 "15|spx_b2.hip|s/lo = SQ\[k\] - lsv; hi = SQ\[k\] + lsv;/lo = SQ[k] - lsv; hi = SQ[k] - lsv;/|tests/test_gpu_stress.py::test_b2_streaming_form_scenarios tests/test_gpu_stress.py::test_b2_one_launch_forms_at_their_boundaries"
 "16|spx_group_common.hpp|0,/if constexpr (TEAM >= 2) v += dpp_f64<0xB1>(v);/s//if constexpr (TEAM >= 4) v += dpp_f64<0xB1>(v);/|tests/test_gpu_parity.py::test_group_uniform"
 "17|spx_objective.hip|s/for (int off = TEAM \/ 2; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);  \/\/ (inside the team/for (int off = TEAM \/ 2; off >= 2; off >>= 1) ss += __shfl_xor(ss, off, 64);  \/\/ (inside the team/|tests/test_gpu_parity.py::test_objective_group_sizes"
 "18|spx_separable.hip|s/^        st2<NT>(y + i, r);\$/        if (blockIdx.x != gridDim.x \/ 2) st2<NT>(y + i, r);/|tests/test_gpu_fullsize.py::test_full_size_lhalf"
 "19|spx_group.hip|s/^    if (valid) {\$/    if (valid \&\& blockIdx.x != gridDim.x \/ 2) {/|tests/test_gpu_fullsize.py::test_full_size_groups"
 "20|spx_group_team.hip|s/        visit(i < npairs, i, qa, xa, sa);/        visit(i < npairs \&\& !(wl == W \/ 2 \&\& tile == wl + W), i, qa, xa, sa);/|tests/test_gpu_team.py::test_one_group_over_the_vector tests/test_gpu_fullsize.py::test_one_group_over_1e8_elements"
 "21|spx_common.hpp|s/  __hip_atomic_store(\&hdr->fin_top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);/  (void)0;/|tests/test_gpu_launch_counts.py::test_objective_one_launch_same_bits"
 "22|spx_objective.hip|s/    __hip_atomic_store(\&fin.hdr->fin_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);/    (void)0;/|tests/test_gpu_launch_counts.py::test_objective_one_launch_same_bits tests/test_gpu_launch_counts.py::test_objective_one_launch_into_a_device_double_back_to_back"
 "23|spx_group.hip|s/if (LIT \&\& dclear != nullptr \&\& blockIdx.x == 0 \&\& threadIdx.x == 0) \*dclear = 0ull;/(void)dclear;/|tests/test_gpu_launch_counts.py::test_binf_deferred_list_without_the_zero_launch"
 "24|spx_separable.hip|s/value_reduce_small<true>(op.partials, (int)gridDim.x);/value_reduce_small<true>(op.partials, (int)gridDim.x - 1);/|tests/test_gpu_launch_counts.py::test_prox_value_one_launch"
 "25|spx_select.hip|s/          for (int e = 0; e < W; ++e) vvv\[e\] = vr\[(s_ >= kSlots) ? s_ - kSlots + e : 0\];/          for (int e = 0; e < W; ++e) vvv[e] = vr[(s_ >= kSlots) ? s_ - kSlots : 0];/|tests/test_gpu_parity.py::test_indball_l0_at_the_fast_path_threshold"
 "26|spx_objective.hip|s/    for (int64_t c = c0 + t; c < c1; c += 256) gs += spx_atomic_load_f64(chunk_ss + c);/    for (int64_t c = c0 + t; c < c1; c += 512) gs += spx_atomic_load_f64(chunk_ss + c);/|tests/test_gpu_launch_counts.py::test_objective_of_large_groups_one_launch"
 "4|spx_group_common.hpp|s/if (sb == 0.0) {/if (false) {/;s/for (int k = 0; k < 64; ++k) {/for (int k = 0; k < 12; ++k) { piece_ok = true;/|tests/test_gpu_parity.py::test_group_binf_many_small_groups tests/test_gpu_parity.py::test_group_binf_zero_groups_strong_lambda"
)
case "${1:-}" in
build)
  only=" ${*:2} "
  for f in "${FAULTS[@]}"; do
    IFS='|' read -r id file expr tests <<< "$f"
    [ "$only" = "  " ] || [[ "$only" == *" $id "* ]] || continue
    W="/tmp/spx_fault_$id"; rm -rf "$W"; mkdir -p "$W/shiftedproximaloperators.jl_amd" "$W/include"
    cp -r "$CSRC" "$W/shiftedproximaloperators.jl_amd/csrc"; cp "$ROOT/include/spx.h" "$W/include/"
    before=$(md5sum < "$W/shiftedproximaloperators.jl_amd/csrc/$file")
    sed -i "$expr" "$W/shiftedproximaloperators.jl_amd/csrc/$file"
    [ "$before" != "$(md5sum < "$W/shiftedproximaloperators.jl_amd/csrc/$file")" ] || { echo "fault $id: the pattern no longer matches $file -- update tools/planted_faults.sh"; exit 1; }
    SPX_LIB_NAME="libspx_fault$id.so" bash "$W/shiftedproximaloperators.jl_amd/csrc/build.sh" > /dev/null || exit 1
    cp "$W/shiftedproximaloperators.jl_amd/lib/libspx_fault$id.so" "$ROOT/shiftedproximaloperators.jl_amd/lib/"
    echo "built libspx_fault$id.so"
  done ;;
run)
  cd "$ROOT"; mkdir -p gpurun_out; bad=0
  only=" ${*:2} "   # (run [ids]: all, or the ids listed)
  for f in "${FAULTS[@]}"; do
    IFS='|' read -r id file expr tests <<< "$f"
    [ "$only" = "  " ] || [[ "$only" == *" $id "* ]] || continue
    SPX_LIB_NAME="libspx_fault$id.so" SPX_NO_BUILD=1 timeout -k 10 300 python -m pytest $tests -m gpu -q -x -p no:cacheprovider > "gpurun_out/fault$id.log" 2>&1
    rc=$?
    nfail=$(grep -c '^FAILED' "gpurun_out/fault$id.log"); nerr=$(grep -c '^ERROR' "gpurun_out/fault$id.log")
    if [ $rc -eq 1 ] && [ "$nfail" -ge 1 ] && [ "$nerr" -eq 0 ]; then echo "fault $id: caught ($nfail failing test(s), first: $(grep -m1 '^FAILED' gpurun_out/fault$id.log | cut -c1-120))"
    elif [ "$nerr" -ge 1 ]; then echo "fault $id: the test run ERRORED (stale library? rebuild with tools/planted_faults.sh build): $(grep -m1 '^ERROR' gpurun_out/fault$id.log | cut -c1-120)"; bad=1
    else echo "fault $id: NOT CAUGHT (pytest rc $rc)"; bad=1; fi
  done
  exit $bad ;;
*) echo "usage: $0 build|run"; exit 2 ;;
esac
