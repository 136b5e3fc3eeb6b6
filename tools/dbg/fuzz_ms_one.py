"""Re-run single trials of tests/fuzz/fuzz_multistep.py:  python tools/dbg/fuzz_ms_one.py SEED T [T ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
seed = int(sys.argv[1])
ts = [int(x) for x in sys.argv[2:]]
sys.argv = [""]
import tests.fuzz.fuzz_multistep as fm
for t in ts:
    fm.one(t, np.random.default_rng([seed, t]))
