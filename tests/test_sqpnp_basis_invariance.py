"""SQPnP on degenerate scenes: does the returned pose depend on choices a linear-algebra library is free to make?

The reference takes Omega's eigenvectors from nalgebra's `symmetric_eigen` and the start rotations from nalgebra's `svd`
(crates/chalkydri_sqpnp/src/lib.rs:45,398).  When every tag sits on one axis-aligned wall — the layout of bench.py's scenes
and of a real FRC field wall — the centred world points have one coordinate exactly zero, Omega has an exact 3-dimensional
null space, and the first eigen-guesses are rank-1 matrices: WHICH orthonormal basis of the null space and WHICH completion
of the rank-1 SVD comes back is implementation-defined, and the oracle (oracle/sqpnp.c: cyclic Jacobi + a deterministic
completion) cannot claim to reproduce nalgebra's.  What can be tested is that it does not matter: tests/np_sqpnp.py draws
all those choices at random (bases of degenerate eigenspaces, order of equal eigenvalues, eigenvector signs, SVD
completions) and the pose must come back the same.

Findings this test pins:
  * on wall scenes with detector-level corner noise and on generic scenes, whenever a pose is returned it equals the oracle's
    to 1e-9 (1e-8 on generic noisy scenes);
  * on NOISE-FREE one-wall scenes the reference's algorithm has no basis-independent answer (second test below);
  * on a one-wall scene a small share of the draws (about 1 %) returns NO pose: all six starts then converge to the mirror
    solution of the planar scene (same energy, points behind the camera) and the cheirality test rejects them, which is
    lib.rs:270-292 returning None.  So validity — not the pose — of an exactly degenerate scene is basis-dependent in the
    reference itself; the oracle's deterministic completion finds the pose in every scene tested here.
"""
import numpy as np
import pytest

import np_sqpnp as N
from test_sqpnp_oracle import _wall_scene

TOL = 1e-9


def _dev(a, b):
    return max(np.abs(a["pos"] - b["pos"]).max(), np.abs(a["rot"] - b["rot"]).max(), abs(a["yaw"] - b["yaw"]))


def test_noisy_wall_scene_pose_does_not_depend_on_the_eigen_basis(oracle):
    noise_px = 0.15   # what a detector's corners carry; bench.py's scenes are rendered frames, i.e. this regime
    draws = none = 0
    worst = 0.0
    for seed in range(10):
        rng = np.random.default_rng(100 + seed)
        tags, b, rtc, truth = _wall_scene(rng, 6, noise_px=noise_px)
        gyro = truth["yaw"] + rng.uniform(-0.02, 0.02)
        want = oracle.sqpnp_solve(tags, b, rtc, gyro)
        assert want is not None
        plain = N.solve_robot_pose(tags, b, rtc, gyro)          # numpy's own eigh / svd choices: the cross-check on wall scenes
        assert plain is not None and _dev(plain, want) < TOL
        for k in range(24):
            got = N.solve_robot_pose(tags, b, rtc, gyro, rng_basis=np.random.default_rng(1000 * seed + k))
            draws += 1
            if got is None:
                none += 1
                continue
            worst = max(worst, _dev(got, want))
    assert worst < TOL, worst
    assert none <= 0.05 * draws, (none, draws)   # the mirror-solution draws (see the module docstring)
    print(f"noisy wall scenes: {draws} draws, worst deviation {worst:.2e}, {none} without a pose")


def test_noise_free_wall_scene_is_ill_conditioned_in_the_reference_algorithm(oracle):
    """Exact data on one wall: the energy is zero at the solution and nearly zero along Omega's weakest non-null direction
    (eigenvalue about 3e-6), the refinement has 15 steps and no re-projection (lib.rs:203-204,463-480) — so where it ends depends
    on the start, i.e. on the eigen-basis: draws end millimetres to metres away from each other, every one with an energy the
    algorithm itself calls a perfect fit.  No basis-independent answer exists for the reference here; the oracle's deterministic
    choice recovers the true pose, which is the only claim made (and why pose parity is asserted on detector-noise inputs)."""
    far = draws = 0
    for seed in range(10):
        rng = np.random.default_rng(100 + seed)
        tags, b, rtc, truth = _wall_scene(rng, 6, noise_px=0.0)
        gyro = truth["yaw"] + rng.uniform(-0.02, 0.02)
        want = oracle.sqpnp_solve(tags, b, rtc, gyro)
        assert want is not None and abs(want["energy"]) < 1e-12
        assert np.abs(want["pos"][:2] - truth["twr"][:2]).max() < 1e-4 + 10 * abs(gyro - truth["yaw"]) ** 3 + 0.03
        for k in range(12):
            got = N.solve_robot_pose(tags, b, rtc, gyro, rng_basis=np.random.default_rng(1000 * seed + k))
            draws += 1
            if got is None:
                continue
            assert abs(got["energy"]) < 1e-6          # a "perfect fit" by the algorithm's own measure (rms < 1e-3 * MAX_TRUSTABLE_RMS)
            far += _dev(got, want) > 1e-3
    print(f"noise-free wall scenes: {far} of {draws} random-basis draws end more than 1e-3 away from the oracle's pose")


def test_generic_scene_pose_does_not_depend_on_signs_and_order(oracle):
    worst = 0.0
    for seed in range(8):
        rng = np.random.default_rng(seed)
        tags, b, rtc, truth = N.make_scene(rng, int(rng.integers(2, 7)), noise_px=0.3 if seed % 2 else 0.0)
        gyro = truth["yaw"] + rng.uniform(-0.1, 0.1)
        want = oracle.sqpnp_solve(tags, b, rtc, gyro)
        assert want is not None
        for k in range(12):
            got = N.solve_robot_pose(tags, b, rtc, gyro, rng_basis=np.random.default_rng(77 * seed + k))
            assert got is not None
            worst = max(worst, _dev(got, want))
    assert worst < 1e-8, worst   # noisy generic scenes: the six refinements stop at tol 1e-8 per step (lib.rs:204)
