"""The smoothed outputs against something that does not move when the kernels do.

tests/golden/ref_*.npz hold the reference's formulas evaluated in 160-digit arithmetic (oracle/referee_mp.py) and the
LAPACK reading (oracle/ekf_numpy.py), both FROZEN, with per-output gates derived from how far the two fp64 readings were
from the exact result when the fixtures were made (tests/golden/make_golden_referee.py).  The C oracle -- which the HIP
kernels must equal bit for bit (tests/test_gpu_parity.py) -- is held to those gates here; the HIP output itself is held to
them in tests/test_gpu_parity.py::test_hip_against_frozen_referee_and_lapack_vectors.  A pinv change that is wrong but
self-consistent between kernel and C oracle fails these tests; regenerating tests/golden/sia*.npz does not help it."""
import numpy as np
import pytest

from tests import helpers as H


@pytest.mark.parametrize("name", H.REFEREE_CASES)
def test_c_oracle_against_frozen_referee_and_lapack_vectors(name):
    w, fx = H.load_referee(name)
    got = H.oracle_batch(w)
    fails, rep = H.referee_compare(w, got, fx, name)
    assert not fails, (fails, rep)
    # the C oracle has not drifted from where it stood when the fixture was frozen by more than the gate allows either:
    # its distance from the exact smoothed epidemic states stays within 10 x the frozen one (or 1e-12)
    if "S_SMOOTH_states" in rep:
        assert rep["S_SMOOTH_states"]["vs_exact"] <= max(1e-12, 10.0 * float(fx["dist_C_S_SMOOTH_states"])), rep["S_SMOOTH_states"]


@pytest.mark.parametrize("name", H.REFEREE_CASES)
def test_lapack_reading_still_gives_its_frozen_vectors(name):
    """oracle/ekf_numpy.py re-run on the fixture's inputs against its own frozen outputs: the LAPACK reading is what the
    GPU reports are written against, so it must not drift silently either (another LAPACK build may move the last digits
    of ill-conditioned quantities: the comparison uses the fixture's gate, and exact equality of the truncation ranks)."""
    w, fx = H.load_referee(name)
    m = w.m
    rows = lambda a: a[None] if a.ndim == 1 else a.reshape(-1, a.shape[-1])
    for c in range(w.B):
        nd = H.numpy_chain(w, c)
        for key in [k for k in fx if k.startswith("lap_")]:
            n = key[4:]
            if n == "pinv_rank":
                assert np.array_equal(nd["pinv_rank"], fx[key][:, c]), (name, c)
                continue
            base = n[:-5] if n.endswith("_diag") else n
            v = np.asarray(nd[base])
            if n.endswith("_diag"):
                v = np.stack([v[i, i] for i in range(m)])
            tol = float(fx["tol_" + base])
            if tol > H.REFEREE_GATE_CAP:
                continue
            e = H.rowwise_abs_rel_err(rows(v), rows(fx[key][..., c]))
            assert e <= tol, (name, c, n, e, tol)


def test_referee_reproduces_its_frozen_output():
    """The referee itself, re-run on one chain of the shortest fixture: deterministic multi-precision arithmetic, so every
    rounded output must come back bit for bit (guards referee_mp.py against accidental edits)."""
    from oracle import referee_mp as rf
    w, fx = H.load_referee("ref_sia6_backward_40")
    c = 1
    r = rf.run_model(w.model, *H.chain_args(w, c))
    for key in [k for k in fx if k.startswith("ref_") and k not in ("ref_near_cutoff",)]:
        n = key[4:]
        assert np.array_equal(np.asarray(r[n]), fx[key][..., c], equal_nan=True), n


def test_referee_precision_is_sufficient():
    """Doubling the working precision must not move the rounded result: 160 digits against 320 on a short 6-state chain
    whose covariances already span 30 orders of magnitude."""
    from oracle import referee_mp as rf
    from epidemicmodeling_amd import synth
    w = synth.make_cfg4(2, 3, 30, 12)
    a = rf.run_model(w.model, *H.chain_args(w, 4))
    b = rf.run_model(w.model, *H.chain_args(w, 4), dps=320)
    for n in ("S_SMOOTH", "P_SMOOTH", "u_opt_smooth", "S_PLUS", "pinv_rank"):
        assert np.array_equal(a[n], b[n], equal_nan=True), n


def test_referee_pinv_known_answers():
    """pinv_exact on prescribed spectra straddling MATLAB's cut-off max(size) * eps(norm): rank, the Moore-Penrose
    identities to ~150 digits, and the near-cut-off ratio it reports."""
    from mpmath import mp, mpf
    from oracle import referee_mp as rf
    old = mp.dps
    mp.dps = 160
    try:
        rng = np.random.default_rng(0)
        Q, _ = np.linalg.qr(rng.standard_normal((6, 6)))
        Qm = [[mpf(float(v)) for v in row] for row in Q]
        # re-orthonormalise in multi-precision (Gram-Schmidt) so that the prescribed values ARE the singular values
        for i in range(6):
            for j in range(i):
                d = mp.fsum(Qm[i][k] * Qm[j][k] for k in range(6))
                Qm[i] = [Qm[i][k] - d * Qm[j][k] for k in range(6)]
            nrm = mp.sqrt(mp.fsum(v * v for v in Qm[i]))
            Qm[i] = [v / nrm for v in Qm[i]]
        tol = 6 * mpf(2) ** -52                       # sigma_max = 1 => eps(1) = 2^-52
        spectra = [([1, 1e-3, 1e-6, 1e-9, 1e-12, 1e-14], 6), ([1, 0.5, float(tol * 3), float(tol / 3), 1e-20, 0.0], 3),
                   ([1, 1e-30, 0, 0, 0, 0], 1)]
        for sv, rank in spectra:
            A = [[mp.fsum(Qm[k][i] * mpf(sv[k]) * Qm[k][j] for k in range(6)) for j in range(6)] for i in range(6)]
            X, r, s, t, ratio = rf.pinv_exact(A)
            assert r == rank, (sv, r)
            AXA = rf._matmul(rf._matmul(A, X), A)
            # A X A = A restricted to the kept subspace: compare with the rank-r truncation of A
            Ar = [[mp.fsum(Qm[k][i] * mpf(sv[k]) * Qm[k][j] for k in range(rank)) for j in range(6)] for i in range(6)]
            err = max(abs(AXA[i][j] - Ar[i][j]) for i in range(6) for j in range(6))
            assert err < mpf(10) ** -100, err     # Q is orthonormal to ~160 digits, so A X A is exactly the kept part of A
            if rank == 3:
                assert 2.9 < float(ratio) < 3.1
    finally:
        mp.dps = old


def test_gates_catch_a_wrong_but_self_consistent_smoother():
    """What the frozen vectors are for: mutations of the kind a mis-used pinv produces (X off by a factor 1 + 1e-3; the
    smoother gain's correction dropped) applied to the C oracle's output must fail the gates, in most fixtures."""
    caught_scale = caught_drop = 0
    for name in H.REFEREE_CASES:
        w, fx = H.load_referee(name)
        got = H.oracle_batch(w)
        bad = dict(got)
        # J scaled by (1 + 1e-3): the smoothed correction S_SMOOTH - S_PLUS grows by that factor
        bad["S_SMOOTH"] = got["S_PLUS"] + (got["S_SMOOTH"] - got["S_PLUS"]) * (1.0 + 1e-3)
        f1, _ = H.referee_compare(w, bad, fx, name)
        caught_scale += bool(f1)
        bad["S_SMOOTH"] = got["S_PLUS"].copy()            # J = 0: smoothed = filtered
        f2, _ = H.referee_compare(w, bad, fx, name)
        caught_drop += bool(f2)
    assert caught_drop == len(H.REFEREE_CASES), caught_drop
    assert caught_scale >= 5, caught_scale


def test_inner_gate_catches_what_the_lapack_derived_gate_lets_through():
    """Round 5: a smoother that loses five digits -- here the smoothed epidemic states off by a relative 1e-10, far inside the
    frozen gate that LAPACK's error sets (2.7e-8 on this fixture) -- fails the inner gate, 100 x the distance the C oracle
    stood at from the exact result when the fixture was frozen (4.5e-13)."""
    name = "ref_cfg4_dead_400_120"
    w, fx = H.load_referee(name)
    got = H.oracle_batch(w)
    assert float(fx["tol_S_SMOOTH_states"]) > 1e-8 and H.referee_inner_gate(fx, "S_SMOOTH_states") < 1e-10
    ok, _ = H.referee_compare(w, got, fx, name)
    assert not ok
    bad = dict(got)
    bad["S_SMOOTH"] = got["S_SMOOTH"].copy()
    bad["S_SMOOTH"][:, :3] *= 1.0 + 1e-10
    fails, _ = H.referee_compare(w, bad, fx, name)
    assert fails and all("inner gate" in f[2] for f in fails), fails
