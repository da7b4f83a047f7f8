"""CPU: bench.py's host-side helpers (argument contract, CPU-set handling) -- no GPU, no timing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_defaults_follow_the_driver_contract(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps is None and a.warmup >= 0 and a.pairs == 271 and a.azimuth * 64 == 65536   # no --steps: timed region sized to >= 0.5 s
    assert a.config4_pairs == 4541                                       # BASELINE config 4 rides along with the default line
    assert a.mode == "direct7" and a.variant == "omp" and a.resolution == 1.0 and a.total_pairs == 0
    assert a.host_clouds is False and a.no_host_clouds is False          # the leg is on by default at N = 1 (main decides)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2", "--total-pairs", "4541"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup, a.total_pairs) == (8, 5, 2, 4541)


def test_allowed_cpu_set_is_captured_and_reapplied():
    """bench.py remembers the CPUs the process may use before numpy / torch (and their OpenMP runtimes) are imported and gives
    them back to the threads it starts itself: a runtime started with OMP_PROC_BIND narrows the loading thread's affinity."""
    import bench
    assert bench.ALLOWED_CPUS and bench.ALLOWED_CPUS <= set(range(os.cpu_count() or 4096))
    before = os.sched_getaffinity(0)
    try:
        one = {sorted(bench.ALLOWED_CPUS)[0]}
        os.sched_setaffinity(0, one)                                     # what an OpenMP runtime may have done to this thread
        n = bench.apply_affinity()
        assert n == len(bench.PIN_CPUS or bench.ALLOWED_CPUS) and n >= 1
        assert os.sched_getaffinity(0) == (bench.PIN_CPUS or bench.ALLOWED_CPUS)
    finally:
        os.sched_setaffinity(0, before)


def test_cpu_quota_is_a_positive_count_or_unknown():
    import bench
    q = bench.cpu_quota()
    assert q is None or q >= 1


def test_parity_sample_order_spans_the_index_range():
    """A time-bounded parity sample has to cover the job's whole index range (config 4: pairs 271..4540 too), not its first pairs."""
    import bench
    for n in (1, 2, 3, 7, 271, 4541):
        o = bench.spread_order(n)
        assert sorted(o) == list(range(n))
        if n > 2:
            assert o[0] == 0 and o[1] == n - 1
    o = bench.spread_order(4541)
    first = sorted(o[:285])
    gaps = [b - a for a, b in zip(first, first[1:])]
    assert max(gaps) <= 32 and first[-1] == 4540                         # ~every 16th pair, no hole wider than 32


def test_workload_label_names_the_baseline_config_only_when_it_is_that_config(monkeypatch):
    import bench
    def args(*extra):
        monkeypatch.setattr(sys, "argv", ["bench.py", *extra])
        return bench.parse()
    assert bench.baseline_config_name(args(), 65536) == "BASELINE config 3"
    assert "another batch size" in bench.baseline_config_name(args("--pairs", "1536"), 65536)
    a5 = args("--variant", "pca", "--resolution", "0.5", "--azimuth", "2048", "--pairs", "128")
    assert bench.baseline_config_name(a5, 131072) == "BASELINE config 5's per-GPU share"
    a5d1 = args("--variant", "pca", "--mode", "direct1", "--resolution", "0.5", "--azimuth", "2048", "--pairs", "128")
    assert "DIRECT1" in bench.baseline_config_name(a5d1, 131072)
    assert bench.baseline_config_name(args("--variant", "pca", "--mode", "direct1"), 65536) == "a variation of BASELINE config 3"
