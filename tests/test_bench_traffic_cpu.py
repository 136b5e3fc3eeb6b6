"""bench.py's own HBM-traffic figure (roofline.traffic): the condensation of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter
files into bytes per launch -- units (KiB), the gfx950 doubling of the read side, the dropped first launch, the variants that
run once per step summed and a one-off variant left out -- on synthetic counter files (no GPU, no profiler)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _write(dirname, ctr, rows):
    d = os.path.join(dirname, "1234")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "1234_counter_collection.csv"), "w") as f:
        f.write('"Correlation_Id","Dispatch_Id","Agent_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
        for k, (name, val) in enumerate(rows):
            f.write('%d,%d,"Agent 1","%s","%s",%r\n' % (k, k, name, ctr, val))


def test_counter_files_to_bytes_per_launch(tmp_path):
    b = _bench()
    fast = "void k_sph_force<10, 1>(SphDev, double const*, int)"
    slow = "void k_sph_force<10, 0>(SphDev, double const*, int)"
    once = "void k_sph_force<10, 3>(SphDev, double const*, int)"
    other = "void k_scatter_adv<false>(AdvanceArgs)"
    fetch = [(once, 9e6)] + [(fast, v) for v in (5e6, 4e6, 4e6, 4e6)] + [(slow, v) for v in (3e4, 2e4, 2e4, 2e4)] + [(other, 7e6)] * 4
    write = [(once, 9e6)] + [(fast, v) for v in (6e6, 5e6, 5e6, 5e6)] + [(slow, v) for v in (1e4, 1e4, 1e4, 1e4)] + [(other, 7e6)] * 4
    _write(str(tmp_path / "FETCH_SIZE"), "FETCH_SIZE", fetch)
    _write(str(tmp_path / "WRITE_SIZE"), "WRITE_SIZE", write)
    per = {c: b._counter_rows(str(tmp_path / c), c, "k_sph_force") for c in ("FETCH_SIZE", "WRITE_SIZE")}
    assert set(per["FETCH_SIZE"]) == {"k_sph_force<10, 1>", "k_sph_force<10, 0>", "k_sph_force<10, 3>"}      # not the scatter pass
    total, parts, top = b._traffic_from_counters(per)
    assert top == 4
    assert "k_sph_force<10, 3>" not in parts                      # one launch only: not the steady state
    want = 1024.0 * (2.0 * (4e6 + 2e4) + (5e6 + 1e4))             # first launch dropped, reads doubled
    assert abs(total - want) < 1e-6 * want
    assert parts["k_sph_force<10, 1>"]["FETCH_SIZE"] == 2.0 * 4e6 * 1024.0


def test_live_traffic_declines_under_a_profiler(monkeypatch):
    b = _bench()
    monkeypatch.setenv("ROCPROFILER_TEST_MARK", "1")
    monkeypatch.setattr("shutil.which", lambda name: "/usr/bin/true")

    class A:
        nbodies, lmax, nmax, numr, dt = 1e4, 6, 18, 2000, 0.002
    val, why = b.live_traffic(A(), "k_sph_force", 10000)
    assert val is None and "profiler" in why
