"""The host half of SphericalBasis's NOISE mode without a GPU: compute_rms_coefs (src/SphericalBasis.cc:2108-2147) of
exp_amd/slgrid.py against the oracle's restatement on the reference's own model table (tests/golden/SLGridSph.model =
tests/Halo/SLGridSph.model), the table reader and interpolation against known answers, and the deviate sequence of the
oracle's update_noise -- std::mt19937 + std::normal_distribution, the objects the reference holds -- against a pure-Python
restatement of libstdc++'s algorithm (Marsaglia's polar method on generate_canonical<double, 53>)."""
import math
import os

import numpy as np
import pytest

from tests.golden_util import load_sph

MODEL = os.path.join(os.path.dirname(__file__), "golden", "SLGridSph.model")


def test_model_table_reader_and_density():
    from exp_amd.slgrid import model_density, read_model_table
    r, d, m, p = read_model_table(MODEL)
    assert len(r) == len(d) == len(m) == len(p) > 100 and np.all(np.diff(r) > 0)
    # on a node, between two nodes (linear: odd2), below the table (extrapolated from the first interval), beyond it (last value)
    assert model_density(r, d, float(r[7])) == pytest.approx(d[7], rel=1e-15)
    x = 0.25 * r[10] + 0.75 * r[11]
    assert model_density(r, d, x) == pytest.approx(0.25 * d[10] + 0.75 * d[11], rel=1e-13)
    assert model_density(r, d, 0.5 * r[0]) == pytest.approx(d[0] + (d[1] - d[0]) * (0.5 * r[0] - r[0]) / (r[1] - r[0]), rel=1e-13)
    assert model_density(r, d, 2.0 * r[-1]) == d[-1]


def test_compute_rms_coefs_against_the_oracle(oracle):
    from exp_amd.slgrid import compute_rms_coefs, read_model_table
    g, _ = load_sph()
    r, d, _, _ = read_model_table(MODEL)
    for scale in (1.0, 0.37):
        meanC, rmsC = compute_rms_coefs(g, MODEL, scale)
        m_ref, r_ref = oracle.sph_compute_rms_coefs(g, r, d, scale)
        assert np.abs(meanC - m_ref).max() <= 1e-13 * np.abs(m_ref).max()
        assert np.abs(rmsC - r_ref).max() <= 1e-13 * np.abs(r_ref).max()
        assert np.all(rmsC > 0)          # (every entry integrates a square; rmsC - meanC^2 may have either sign -- the table's
        #                                   mass is not 1 -- and update_noise takes its absolute value, :2197)


class _StdNormal:
    """libstdc++'s std::normal_distribution<double>(0, 1) on std::mt19937 (bits/random.tcc: Marsaglia's polar method; the
    second deviate of a pair is kept for the next call; generate_canonical<double, 53> takes two 32-bit words, low first)."""

    def __init__(self, seed):
        self.bits = np.random.MT19937()
        # std::mt19937::seed(value): the Knuth initialisation numpy calls `_legacy_seeding`
        self.bits._legacy_seeding(int(seed))
        self.saved = None

    def _u32(self):
        return int(self.bits.random_raw())

    def _canonical(self):
        lo, hi = self._u32(), self._u32()
        v = (lo + hi * 4294967296.0) / 18446744073709551616.0
        return math.nextafter(1.0, 0.0) if v >= 1.0 else v

    def __call__(self):
        if self.saved is not None:
            v, self.saved = self.saved, None
            return v
        while True:
            x = 2.0 * self._canonical() - 1.0
            y = 2.0 * self._canonical() - 1.0
            r2 = x * x + y * y
            if not (r2 > 1.0 or r2 == 0.0):
                break
        mult = math.sqrt(-2.0 * math.log(r2) / r2)
        self.saved = x * mult
        return y * mult


@pytest.mark.parametrize("seed", [11, 123456789])
def test_update_noise_draws_are_the_standard_library_sequence(oracle, seed):
    """update_noise (src/SphericalBasis.cc:2150-2210): row by row, cos then sin per n for m > 0, meanC added on l = 0, and the
    generator seeded ONCE -- two calls continue the sequence"""
    lmax, nmax = 3, 4
    rng = np.random.default_rng(5)
    meanC = rng.normal(0, 1, nmax)
    rmsC = meanC[None, :] ** 2 + rng.uniform(0.1, 2.0, (lmax + 1, nmax))
    noiseN = 1.0e-6
    h = oracle.noise_create(lmax, nmax, meanC, rmsC, noiseN, seed)
    got = [oracle.noise_update(h, lmax, nmax) for _ in range(2)]
    oracle.lib.orc_noise_destroy(h)
    nr = _StdNormal(seed)
    for call in range(2):
        ref = np.zeros(((lmax + 1) ** 2, nmax))
        for l in range(lmax + 1):
            for m in range(l + 1):
                fac = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m)) * (math.sqrt(2.0) if m else 1.0)
                amp = np.sqrt(np.abs(rmsC[l] - meanC ** 2) * fac / noiseN)
                row = l * l + (0 if m == 0 else 2 * m - 1)
                for n in range(nmax):
                    if m == 0:
                        ref[row, n] = amp[n] * nr() + (meanC[n] if l == 0 else 0.0)
                    else:
                        ref[row, n] = amp[n] * nr()
                        ref[row + 1, n] = amp[n] * nr()
        assert np.abs(got[call] - ref).max() <= 1e-12 * np.abs(ref).max(), call
    assert np.abs(got[0] - got[1]).max() > 0.1 * np.abs(got[0]).max()
