"""The two host-side campaigns of tests/fuzz/ (coefficient containers and their files, phase-space files), a bounded
fixed-seed slice each, inside the CPU suite: the first trials of the committed campaign logs, values bit for bit."""
import importlib

import numpy as np

SEED = 1


def test_fuzz_coefs_slice():
    m = importlib.import_module("tests.fuzz.fuzz_coefs")
    bad = [t for t in range(40) if not m.one(t, np.random.default_rng([SEED, t]))]
    assert not bad, f"fuzz_coefs mismatches at seed {SEED}: trials {bad}"


def test_fuzz_reader_slice():
    m = importlib.import_module("tests.fuzz.fuzz_reader")
    bad = [t for t in range(40) if not m.one(t, np.random.default_rng([SEED, t]))]
    assert not bad, f"fuzz_reader mismatches at seed {SEED}: trials {bad}"
