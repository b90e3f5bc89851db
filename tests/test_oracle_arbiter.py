"""The binary128 arbiter (oracle/spx_oracle_q.c) pinned the same way as the Float64 oracle: against the reference's own
golden vectors, against the Float64 oracle on well-conditioned data, and the arbiter inequality of tests/arbiter.py
against planted faults (it must reject a result that is further from the exact value than the Float64 oracle)."""
import numpy as np
import pytest

import arbiter


def _data(n, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n), rng.normal(size=n)


def test_arbiter_reference_goldens(orc, kats):
    k = kats["box_golden"]                      # test/runtests.jl:449-494
    x, q = np.array(k["x"]), np.array(k["q"])
    y = orc.q_prox_lhalf_box(q, x, np.zeros_like(x), k["lambda"], k["sigma"], -k["delta"], k["delta"])
    np.testing.assert_allclose(y, k["expected"]["ShiftedRootNormLhalfBox"], rtol=k["rtol"], atol=0)
    for name in ("group_l2_binf_single", "group_l2_binf_two"):   # test/runtests.jl:587-606, 658-705
        k = kats[name]
        x, q = np.array(k["x"]), np.array(k["q"])
        off = np.array(k["offsets"])
        y = orc.q_prox_group_l2(q, x, np.zeros_like(x), k["lambda"], k["sigma"], np.arange(len(off) - 1), offsets=off,
                                binf_delta=k["delta"])
        np.testing.assert_allclose(y, k["expected"], rtol=k["rtol"], atol=0)


def test_arbiter_agrees_with_float64_oracle_on_well_conditioned_data(orc):
    n, gs = 128 * 60, 128
    x, sj, q = _data(n, 1)
    lam = np.random.default_rng(2).uniform(0.5, 1.5, size=n // gs)
    offs = np.arange(0, n + 1, gs)
    allg = np.arange(n // gs)
    for delta in (None, 1.0):
        ref = orc.prox_group_l2_binf(q, x, sj, lam, 1.0, delta, gsize=gs) if delta else orc.prox_group_l2(q, x, sj, lam, 1.0, gsize=gs)
        yq = orc.q_prox_group_l2(q, x, sj, lam, 1.0, allg, gsize=gs, binf_delta=delta)
        sc = arbiter.group_scale(ref, q, x, sj, offs)
        assert np.max(np.abs(ref - yq) / sc) <= 1e-14
    ref, yq = orc.prox_lhalf(q, x, sj, 0.7, 1.3), orc.q_prox_lhalf(q, x, sj, 0.7, 1.3)
    assert np.max(np.abs(ref - yq) / arbiter.lhalf_scale(ref, x, sj, q)) <= 1e-14
    # ... while RELATIVE TO |y| the reference's own Float64 evaluation is off by far more than 1e-12 wherever
    # y = val - (xk + sj) cancels: this is why the bar is stated on the operands' scale
    assert np.max(np.abs(ref - yq) / np.maximum(np.abs(yq), 1e-300)) > 1e-13
    l, u = -1.0 - 0.1 * np.abs(x), 1.0 + 0.1 * np.abs(q)
    ref = orc.prox_lhalf_box(q, x, sj, 0.7, 1.3, l, u)
    yq, cand = orc.q_prox_lhalf_box(q, x, sj, 0.7, 1.3, l, u, return_candidate=True)
    assert np.max(np.abs(ref - yq) / arbiter.lhalf_scale(ref, x, sj, q)) <= 1e-14
    assert set(np.unique(cand)) == {0, 1, 2, 3}
    for dl in (0.1, 1.0, 5.0):
        ref, yq = orc.prox_l1_b2(q, x, sj, 1.0, 1.0, dl), orc.q_prox_l1_b2(q, x, sj, 1.0, 1.0, dl)
        assert np.max(np.abs(ref - yq)) <= 1e-14 * np.linalg.norm(ref)


def test_arbiter_inequality_rejects_planted_faults(orc):
    """A 'GPU' result that is the Float64 oracle pushed away from the exact value must fail; one pushed towards it, or
    within the bar, must pass; the ill-conditioned lambda = 1e6 group shows the Float64 oracle's own error (~3e-8 of the scale)."""
    n, gs = 128 * 8, 128
    x, sj, q = _data(n, 5)
    lam = np.full(n // gs, 1.0)
    lam[3] = 1e6
    offs = np.arange(0, n + 1, gs)
    ref = orc.prox_group_l2_binf(q, x, sj, lam, 1.0, 1.0, gsize=gs)
    yq = orc.q_prox_group_l2(q, x, sj, lam, 1.0, np.arange(n // gs), gsize=gs, binf_delta=1.0)
    sc = arbiter.group_scale(ref, q, x, sj, offs)
    k = slice(3 * gs, 4 * gs)
    own = float(np.max(np.abs(ref[k] - yq[k]) / sc[k]))
    assert 1e-14 < own < 1e-5, own                 # the reference formula's own conditioning in that group (measured: 3e-8)
    # towards the exact value: accepted even though it differs from the Float64 oracle by more than the bar
    better = ref.copy()
    better[k] = yq[k] + 0.01 * (ref[k] - yq[k])
    v = arbiter.check_group(orc, better, ref, q, x, sj, lam, 1.0, offs, delta=1.0)
    assert (v.n_checked == 1 and v.gpu_closer == 1) or own <= 1e-12
    # away from the exact value by more than the reference's own ensemble (norms moved by a few ulps) reaches: rejected
    worse = ref.copy()
    worse[k] = ref[k] + np.sign(ref[k] - yq[k] + 1e-300) * 1e-3 * np.abs(ref[k] - yq[k]).max()
    with pytest.raises(AssertionError, match="further from the binary128"):
        arbiter.check_group(orc, worse, ref, q, x, sj, lam, 1.0, offs, delta=1.0)
    # a well-conditioned group off by 5e-12 of its scale: rejected (the ensemble is ~1e-16 wide there)
    worse = ref.copy()
    worse[gs:2 * gs] += 5e-12 * sc[gs:2 * gs]
    with pytest.raises(AssertionError, match="further from the binary128"):
        arbiter.check_group(orc, worse, ref, q, x, sj, lam, 1.0, offs, delta=1.0)
    # the last-ulp noise of alpha = 1 - sigma*lambda/||w|| in the ill-conditioned group (6e-11 absolute, either sign: what
    # the GPU showed against the literal oracle) is inside the ensemble: accepted
    xs = (x + sj)[k]
    aw = ref[k] + xs                                  # alpha * w of that group
    for sgn in (1.0, -1.0):
        noisy = ref.copy()
        noisy[k] = aw * (1.0 + sgn * 4e-12 * sc[k].max() / np.abs(aw).max()) - xs
        v = arbiter.check_group(orc, noisy, ref, q, x, sj, lam, 1.0, offs, delta=1.0)
        assert v.n_checked == 1
    # a mis-decided group (zeros where the reference has a root) is rejected whatever its share of the data
    wrong = ref.copy()
    wrong[:gs] = -(x + sj)[:gs]
    with pytest.raises(AssertionError, match="further from the binary128"):
        arbiter.check_group(orc, wrong, ref, q, x, sj, lam, 1.0, offs, delta=1.0)
    # separable form
    refl = orc.prox_lhalf(q, x, sj, 1.0, 1.0)
    bad = refl.copy()
    bad[17] += 5e-12 * max(abs(refl[17]), abs(x[17] + sj[17]), abs(q[17]))
    with pytest.raises(AssertionError, match="further from the binary128"):
        arbiter.check_lhalf(orc, bad, refl, q, x, sj, 1.0, 1.0)
    assert arbiter.check_lhalf(orc, refl, refl, q, x, sj, 1.0, 1.0).n_checked == 0
