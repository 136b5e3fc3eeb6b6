"""The oracle's Component::fix_positions with the escape (consp / tidal / rcom) and freeze tests of the thread body
(src/Component.cc:3317-3336) against a plain numpy statement of the same loop: what is flagged, when, and what the
sums leave out.  CPU only."""
import numpy as np


def _beyond(pos, com0, ctr, rad):
    d = pos - com0 - ctr
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]) > rad * rad


def test_escape_and_freeze_in_the_oracle(oracle):
    rng = np.random.default_rng(5)
    n, ms = 4000, 2
    m = rng.uniform(0.5, 1.5, n) / n
    pos, vel, acc = rng.standard_normal((3, n, 3))
    lev = rng.integers(0, ms + 1, n).astype(np.int32)
    com0, ctr, rcom, rtrunc = np.array([0.1, 0.0, -0.1]), np.array([0.0, 0.2, 0.0]), 1.5, 2.0
    iattr = np.zeros(n, np.int32)
    sums = np.zeros((ms + 1, 10))
    model_flag = np.zeros(n, bool)
    model_lev = np.zeros((ms + 1, 10))

    def model(p, mlevel):
        nonlocal model_flag
        exam = lev >= mlevel
        newly = exam & _beyond(p, com0, ctr, rcom) & ~model_flag
        model_flag = model_flag | newly
        take = exam & ~model_flag & ~_beyond(p, com0, ctr, rtrunc)
        for L in range(mlevel, ms + 1):
            k = take & (lev == L)
            model_lev[L] = np.concatenate([[m[k].sum()], (m[k, None] * p[k]).sum(0), (m[k, None] * vel[k]).sum(0),
                                           (m[k, None] * acc[k]).sum(0)])
        tot = model_lev.sum(0)
        tot[1:] /= tot[0]
        return tot

    for p, mlevel in ((pos, 0), (pos + 0.4 * vel, 1), (pos + 0.4 * vel, 0), (pos, 0)):
        ref = oracle.fix_positions_opts(m, p, vel, acc, lev, ms, mlevel, sums, com0, ctr, rcom, iattr, rtrunc)
        want = model(p, mlevel)
        assert np.abs(ref - want).max() <= 1e-12 * np.abs(want).max()
        assert np.array_equal(iattr.astype(bool), model_flag)
    assert 0.05 * n < iattr.sum() < 0.9 * n
    # both tests off: the plain function
    s1, s2 = np.zeros((ms + 1, 10)), np.zeros((ms + 1, 10))
    a = oracle.fix_positions_opts(m, pos, vel, acc, lev, ms, 0, s1, com0, ctr)
    b = oracle.fix_positions(m, pos, vel, acc, lev, ms, 0, s2)
    assert np.array_equal(a, b)
