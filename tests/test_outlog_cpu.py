"""The run log's file format (exp_amd/outlog.py against src/OutLog.cc): header lines and data rows character for character
against the oracle's printf restatement, the label tables against the reference's source text, and the reference's own
reader of the file (tests/Halo/check.py) applied to it.  CPU only (the sums themselves are the device's:
tests/test_outlog_gpu.py)."""
import os
import re

import numpy as np
import pytest

from exp_amd import outlog

REF = "/root/reference"


def _sums(rng, n):
    return {"mtot": float(rng.uniform(0.5, 2)), "com": rng.normal(size=3) * 1e-3, "cov": rng.normal(size=3) * 1e-4,
            "angm": rng.normal(size=3), "ektot": float(rng.uniform(0.1, 1)), "eptot": -float(rng.uniform(0.2, 2)),
            "clausius": -float(rng.uniform(0.2, 2)), "nbodies": n}


def test_rows_match_the_oracles_printf(oracle):
    rng = np.random.default_rng(3)
    for ncomp in (1, 2, 3):
        for precision in (10, 4, 16):
            sums = [_sums(rng, int(rng.integers(1, 100000))) for _ in range(ncomp)]
            sums[0]["nbodies"] = 1 if precision == 4 else sums[0]["nbodies"]      # a single body keeps its bulk kinetic energy
            if precision == 16:
                sums[-1].update(mtot=0.0, clausius=0.0)                            # empty component: the guarded divisions
            ctr = rng.normal(size=(ncomp, 3))
            used = [int(s["nbodies"]) - 1 for s in sums]
            got = outlog.row_from_sums(1.234e-2, sums, ctr, used, 0.0123, precision)
            want = oracle.outlog_row(1.234e-2, sums, ctr, used, 0.0123, precision)
            assert got == want
            cols = got.rstrip("\n").split("|")
            assert len(cols) == 19 + 20 * ncomp and all(len(c) == 10 + precision for c in cols)


def test_header_layout_and_the_references_reader(tmp_path):
    """Six header lines, then rows: tests/Halo/check.py skips exactly six lines, splits on '|' and averages the 17th column."""
    rng = np.random.default_rng(5)
    names, ids = ["halo"], ["sphereSL"]
    path = tmp_path / "OUTLOG.run0"
    head = outlog.header(names, ids, 10)
    lines = head.split("\n")
    assert len(lines) == 7 and lines[6] == "" and len({len(l) for l in lines[:6]}) == 1
    assert lines[0].startswith("--------Global stats|") and lines[0].split("|")[19].strip() == "sphereSL"
    assert lines[2].split("|")[16].strip() == "2T/VC" and lines[2].split("|")[19].strip() == "halo mass"
    assert lines[2].split("|")[37].strip() == "halo 2T/VC" and lines[4].split("|")[38].strip() == "[39]"
    assert set(lines[1]) == {"-", "+"} and lines[1] == lines[3] == lines[5]
    want = []
    with open(path, "w") as f:
        f.write(head)
        for k in range(8):
            s = _sums(rng, 10000)
            want.append(-2.0 * s["ektot"] / s["clausius"])
            f.write(outlog.row_from_sums(0.02 * k, [s], [np.zeros(3)], [10000], 0.1))
    # what tests/Halo/check.py does with the file: six header lines skipped, rows split on '|', column 17 averaged
    lines = open(path).read().splitlines()
    vals = [float(ln.split("|")[16]) for ln in lines[6:]]
    assert len(lines) == 14 and np.mean(vals) == pytest.approx(np.mean(want), rel=1e-9)
    # a label longer than the column is written unpadded (src/OutLog.cc:305-309)
    long_head = outlog.header(["a component with a long name"], ["cylinder"], 4).split("\n")
    assert "|a component with a long name 2T/VC|" in long_head[2] and len(long_head[2]) > len(long_head[1])


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference sources")
def test_labels_are_the_references():
    src = open(os.path.join(REF, "src/OutLog.cc")).read()
    g = re.search(r"lab_global\[\]\[19\] = \{(.*?)\};", src, re.S).group(1)
    c = re.search(r"lab_component\[\]\[20\] = \{(.*?)\};", src, re.S).group(1)
    assert re.findall(r'"([^"]+)"', g) == outlog.LAB_GLOBAL and re.findall(r'"([^"]+)"', c) == outlog.LAB_COMPONENT
    hdr = open(os.path.join(REF, "src/OutLog.H")).read()
    assert "num_global = 19" in hdr and "num_component = 20" in hdr
    assert "const int cwid = 10 + precision;" in src and "precision = 10;" in src
    # the statements the row hinges on
    for stmt in ("clausius1[indx] += p->mass*posL[k]*p->acc[k];", "ektot1[indx]    += 0.5*p->mass*velL[k]*velL[k];",
                 "eptot1[indx]  += 0.5*p->mass*p->pot;", "if (nbodies[i]>1) ektot[i] -= 0.5*mtot[i]*vbar2;",
                 'out << "|" << setw(cwid) << -2.0*ektot0/clausius0;'):
        assert stmt in src, stmt


def test_restart_keeps_the_rows_up_to_the_current_time(tmp_path):
    rng = np.random.default_rng(7)
    path = str(tmp_path / "OUTLOG.r")
    with open(path, "w") as f:
        f.write(outlog.header(["halo"], ["sphereSL"]))
        for k in range(6):
            f.write(outlog.row_from_sums(0.1 * k, [_sums(rng, 10)], [np.zeros(3)], [10], 0.0))
    before = open(path).read().split("\n")

    class Store:
        center = np.zeros(3)

        def log_sums(self):
            return _sums(rng, 10)

    class Force:
        def Used(self):
            return 10
    log = outlog.OutLog(path, nint=1, restart=True)
    log.add_component("halo", "sphereSL", Store(), Force())
    log.run(31, 0.31)
    after = open(path).read().split("\n")
    assert os.path.exists(path + ".bak") and open(path + ".bak").read().split("\n") == before
    # the first copy loop stops after the first line holding ANY of the letters T, i, m, e (find_first_of: the first line,
    # through the force id "sphereSL"); the second keeps every line whose leading number is <= tnow -- the rest of the
    # header parses as 0 -- so: header, the rows up to 0.3, the new row
    assert after[:6] == before[:6] and after[6:10] == before[6:10] and len(after) == 6 + 4 + 1 + 1
    assert float(after[10].split("|")[0]) == 0.31
