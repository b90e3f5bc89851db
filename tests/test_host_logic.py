"""Host-side logic of the mirror that needs no GPU: argument validation, group layouts (uniform / CSR / gather),
selection masks and constructor errors on host (numpy) vectors.  Nothing here computes a prox."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def s():
    import __graft_entry__ as ge
    return ge.build()


def test_vector_validation(s):
    import torch
    with pytest.raises(TypeError):
        s.shifted(s.NormL1(1.0), np.ones(4, dtype=np.float32))          # Float32: not on the accelerated path
    with pytest.raises(TypeError):
        s.shifted(s.NormL1(1.0), np.ones(8)[::-1])                       # negative stride: not a view the kernels can take
    base = np.arange(8.0)
    pv = s.shifted(s.NormL1(1.0), base[::2])                             # strided view (the reference's `view(y, 1:2:10)`): packed copy
    assert pv.xk.flags["C_CONTIGUOUS"] and pv.xk.tolist() == [0.0, 2.0, 4.0, 6.0] and pv._xk_ref is not None
    base[2] = -5.0
    pv._refresh()                                                        # every call that reads xk starts with this
    assert pv.xk.tolist() == [0.0, -5.0, 4.0, 6.0]
    s.shift_bang(pv, np.full(4, 9.0))                                    # `ψ.xk .= shift` lands in the caller's array
    assert base.tolist() == [9.0, 1.0, 9.0, 3.0, 9.0, 5.0, 9.0, 7.0]
    om = s.shifted(pv, np.zeros(4))
    assert om._xk_ref is pv._xk_ref and om.xk is pv.xk
    with pytest.raises(TypeError):
        s.shifted(s.NormL1(1.0), torch.ones(4, dtype=torch.float64))     # CPU torch tensor: never staged silently
    with pytest.raises(TypeError):
        s.shifted(s.NormL1(1.0), [1.0, 2.0])
    psi = s.shifted(s.NormL1(1.0), np.ones(4))
    assert psi.host and psi.sol.shape == (4,) and not psi.shifted_twice
    with pytest.raises(IndexError):
        s.shifted(psi, np.ones(5))                                       # BoundsError
    with pytest.raises(TypeError):
        s.shifted(s.IndBallL0(1), np.ones(4), 1.0, s.NormLinf(1.0), [0, 1])   # `selected` only for the Box operators


def test_box_constructor_checks_on_host(s):
    x = np.zeros(3)
    lo, up = np.array([0.0, 1.0, 0.0]), np.array([1.0, 0.5, 1.0])
    for H in (s.NormL1, s.NormL0):                                       # src/shiftedNormL1Box.jl:33-35, L0Box :33-35
        with pytest.raises(ValueError, match="lower bound is greater"):
            s.shifted(H(1.0), x, lo, up)
        with pytest.raises(ValueError):
            s.shifted(H(1.0), x, 2.0, 1.0)
    s.shifted(s.RootNormLhalf(1.0), x, lo, up)                           # no such check (src/shiftedRootNormLhalfBox.jl:22-44)
    psi = s.shifted(s.NormL1(1.0), x, -1.0, 1.0, [2, 0, 2])
    assert psi._mask[0].tolist() == [1, 0, 1]
    assert s.shifted(s.NormL1(1.0), x, -1.0, 1.0, range(0, 3))._mask is None   # selected == 1:n
    om = s.shifted(psi, np.ones(3))
    assert om._mask is psi._mask and om.l == -1.0 and om.shifted_twice
    s.set_bounds_bang(om, lo, 2.0)
    assert om.l is lo and om.u == 2.0
    s.set_radius_bang(om, 0.25)
    assert om.l == -0.25 and om.u == 0.25


def test_group_layouts(s):
    x = np.zeros(12)
    lay = lambda idx, lam=None: s.shifted(s.GroupNormL2(lam or [1.0] * len(idx), idx), x)._layout
    g = lay([range(0, 4), range(4, 8), range(8, 12)])                    # uniform: no index array at all
    assert g.offsets is None and g.group_size == 4 and g.ngroups == 3 and g.index is None
    g = lay([range(0, 3), [3, 4, 5, 6], slice(7, 12)])                   # consecutive ragged ranges: CSR offsets
    assert g.index is None and g.offsets.tolist() == [0, 3, 7, 12] and g.group_size == 5     # size bound (hint)
    g = lay([range(2, 5), range(5, 9)])                                  # consecutive but not spanning 0:n: still CSR
    assert g.index is None and g.offsets.tolist() == [2, 5, 9]
    g = lay([[0, 2, 4], [1, 3, 5]])                                      # true index sets: gather form
    assert g.index.tolist() == [0, 2, 4, 1, 3, 5] and g.offsets.tolist() == [0, 3, 6] and g.nnz == 6
    g = lay([range(4, 8), range(0, 4)])                                  # out of order: gather (sequential semantics)
    assert g.index is not None
    g = lay([range(0, 5), range(3, 8)])                                  # overlapping: gather
    assert g.index is not None and g.nnz == 10
    g = lay([range(0, 12, 2)])                                           # strided range: gather
    assert g.index.tolist() == [0, 2, 4, 6, 8, 10]
    with pytest.raises(IndexError):
        lay([[0, 12]])
    with pytest.raises(IndexError):
        lay([range(8, 13)])
    with pytest.raises(ValueError):
        s.GroupNormL2([1.0, -1.0], [range(0, 6), range(6, 12)])          # src/groupNormL2.jl:20-21
    u = s.shifted(s.GroupNormL2.uniform([0.5, 0.6, 0.7], 4), x)._layout
    assert u.group_size == 4 and u.ngroups == 3 and u.lam.tolist() == [0.5, 0.6, 0.7]
    with pytest.raises(IndexError):
        s.shifted(s.GroupNormL2.uniform([0.5, 0.6], 4), x)
    r = s.shifted(s.GroupNormL2.ragged([1.0, 2.0, 3.0], [0, 5, 6, 12]), x)._layout        # CSR offsets, no Python ranges
    assert r.offsets.tolist() == [0, 5, 6, 12] and r.group_size == 6 and r.ngroups == 3 and r.index is None
    r = s.shifted(s.GroupNormL2.ragged([1.0, 2.0], [0, 6, 12]), x)._layout
    assert r.offsets is None and r.group_size == 6                                         # uniform after all
    with pytest.raises(ValueError):
        s.GroupNormL2.ragged([1.0, 2.0], [0, 7, 5])
    with pytest.raises(IndexError):
        s.shifted(s.GroupNormL2.ragged([1.0], [0, 13]), x)
    one = s.shifted(s.NormL2(0.3), x)                                    # NormL2 -> one group [:]
    assert type(one).__name__ == "ShiftedGroupNormL2" and one._layout.ngroups == 1 and one._layout.group_size == 12


def test_no_gpu_means_loud_failure(s):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    psi = s.shifted(s.NormL1(1.0), np.ones(4))
    with pytest.raises(Exception):
        s.prox(psi, np.ones(4), 1.0)                                     # host vectors are staged through a GPU: none here


def test_build_fails_loudly_on_a_stale_library_without_a_compiler(tmp_path, monkeypatch):
    """build() must rebuild when a source is newer than the shipped libspx.so -- and raise, not run the old
    binary, when there is no hipcc to rebuild with."""
    import os
    import time
    import __graft_entry__ as ge
    lib = os.path.join(ge.PKG, "lib", "libspx.so")
    if not os.path.exists(lib):
        pytest.skip("libspx.so not built yet")
    src = os.path.join(ge.PKG, "csrc", "spx_common.hpp")
    st = os.stat(src)
    monkeypatch.setenv("HIPCC", str(tmp_path / "no_such_hipcc"))
    monkeypatch.delenv("SPX_NO_BUILD", raising=False)
    try:
        os.utime(src, (time.time() + 5, time.time() + 5))      # the source is now newer than the library
        with pytest.raises(RuntimeError, match="no hipcc"):
            ge.build(with_oracle=False)
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    """`--gpus` used to be parsed and ignored.  Under a launcher whose WORLD_SIZE differs it now exits non-zero
    (before anything touches the GPU); without a launcher and --gpus > 1 it starts the ranks itself (GPU box only)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)
