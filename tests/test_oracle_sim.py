"""The simulators' oracle: Philox4x32-10 against the Random123 known-answer vectors, statistics of the normal streams,
and the simulated state-space statistics against the model's closed-form moments."""
import numpy as np
import pytest

from oracle import np_sim, np_models


# kat_vectors of the Random123 distribution (philox4x32 10): counter, key -> output
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize('ctr, key, want', KAT)
def test_philox_known_answers(ctr, key, want):
    got = np_sim.philox4x32_10(np.array(ctr, dtype=np.uint64), key)
    assert tuple(int(v) for v in got) == want


def test_uniform_open_interval_and_exact():
    lo = np_sim.uniform52(np.uint32(0), np.uint32(0))
    hi = np_sim.uniform52(np.uint32(0xFFFFFFFF), np.uint32(0xFFFFFFFF))
    assert lo == 2.0 ** -53 and hi == 1.0 - 2.0 ** -53


def test_normal_streams_moments_and_independence():
    z0, z1 = np_sim.normal_pairs(1234, np.arange(200)[:, None], np.arange(2000)[None, :], np_sim.STREAM_STATE)
    z = np.concatenate([z0.ravel(), z1.ravel()])
    n = z.size
    assert abs(z.mean()) < 4 / np.sqrt(n)
    assert abs(z.var() - 1) < 4 * np.sqrt(2 / n)
    assert abs(np.mean(z ** 4) - 3) < 4 * np.sqrt(96 / n)
    assert abs(np.mean(z0 * z1)) < 4 / np.sqrt(z0.size)
    assert abs(np.mean(z0[:, 1:] * z0[:, :-1])) < 4 / np.sqrt(z0.size)          # along the index
    assert abs(np.mean(z0[1:] * z0[:-1])) < 4 / np.sqrt(z0.size)                # across trials
    y0, _ = np_sim.normal_pairs(1234, np.arange(200)[:, None], np.arange(2000)[None, :], np_sim.STREAM_MEAS)
    assert abs(np.mean(z0 * y0)) < 4 / np.sqrt(z0.size)                         # across streams


def test_simulate_linear_moments():
    """Stationary AR(1)-type model started in its stationary law: every x_k ~ N(0, P_inf); y = H x + noise."""
    F = np.array([[0.9, 0.1], [0.0, 0.8]])
    Sigma = np.array([[0.2, 0.05], [0.05, 0.1]])
    P = np.eye(2)
    for _ in range(500):
        P = F @ P @ F.T + Sigma
    H, Xi = np.array([1.0, -0.5]), 0.3
    B = 2000
    xs, ys = np_sim.simulate(lambda x, dt: (F @ x, Sigma), H, Xi, np.zeros(2), P, 0.1, 12, 99, range(B))
    for k in (0, 7, 11):
        np.testing.assert_allclose(np.cov(xs[:, k].T), P, atol=0.12)
    r = ys - xs @ H
    assert abs(r.var() - Xi) < 0.01
    np.testing.assert_allclose(xs[:, 6].T @ xs[:, 5] / B, F @ P, atol=0.12)     # E[x_{k+1} x_k^T] = F P


def test_simulate_is_shard_invariant_and_seeded():
    cmc = np_models.disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
    H, Xi, m0, P0 = np.array([1.0, 0, 0, 0]), 0.1, np.array([0.0, 0, 7, 0]), np.diag([0.1, 0.1, 1.0, 3.0])
    xa, ya = np_sim.simulate(cmc, H, Xi, m0, P0, 1e-3, 9, 5, range(6))
    xb, yb = np_sim.simulate(cmc, H, Xi, m0, P0, 1e-3, 9, 5, range(3, 6))
    np.testing.assert_array_equal(xa[3:], xb)
    np.testing.assert_array_equal(ya[3:], yb)
    xc, _ = np_sim.simulate(cmc, H, Xi, m0, P0, 1e-3, 9, 6, range(3))
    assert not np.allclose(xa[:3], xc)


def test_add_noise_matches_simulate_measurement_stream():
    F, Sigma = np.eye(1) * 0.5, np.eye(1) * 0.1
    xs, ys = np_sim.simulate(lambda x, dt: (F @ x, Sigma), np.array([1.0]), 0.4, np.zeros(1), np.eye(1), 1.0, 11, 3, range(4))
    ys2 = np_sim.add_noise(xs[:, :, 0], 0.4, 3, range(4), 11)
    np.testing.assert_allclose(ys2, ys, rtol=0, atol=1e-15)
