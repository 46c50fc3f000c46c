"""CPU checks: the committed golden vectors are reproduced by both oracles (NumPy and C port)."""
import pytest

from tests import backends as bk
from tests import cases as cs

CASES = {
    'linear_ou': (lambda: cs.linear_case(0, T=200), 200),
    'chirp_gh3': (lambda: cs.chirp_case(T=200, seed=21), 100),
    'harmonic3_cubature': (lambda: cs.harmonic_case(T=120, seed=22, nh=3), 60),
    'lascala_gh3': (lambda: cs.lascala_case(T=120, seed=23), 60),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_port_reproduces_golden(name):
    make, cd_T = CASES[name]
    c = make()
    z, want = bk.load_golden(name)
    assert (z['ys'] == c.ys).all(), 'seeded inputs changed'
    bk.compare(bk.run_pairs('port', c, cd_T), want, 1e-8, f'port/{name}')


def test_numpy_oracle_reproduces_golden_chirp():
    make, cd_T = CASES['chirp_gh3']
    c = make()
    _, want = bk.load_golden('chirp_gh3')
    bk.compare(bk.run_pairs('numpy', c, cd_T, only=('ekf', 'cd_ekf')), want, 1e-12, 'numpy/chirp_gh3')
