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


def test_gpus_n_without_a_launcher_re_executes_itself_as_n_ranks(monkeypatch):
    """`python bench.py --gpus 8` with no WORLD_SIZE in the environment must become 8 ranks under torch.distributed.run (the driver's own
    launch form, rendezvous on 127.0.0.1) -- never ONE rank that reports n_gpus = 1 -- and must refuse when fewer devices are visible."""
    import bench
    import pytest
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    a = bench.parse()
    seen = {}
    def fake_execve(path, argv, env):
        seen.update(path=path, argv=list(argv), env=dict(env))
    monkeypatch.delenv("LV_SLAM_BENCH_BACKEND", raising=False)
    bench.self_launch(a, sys.argv[1:], visible_devices=8, execve=fake_execve)
    cmd = seen["argv"]
    assert seen["path"] == sys.executable and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]          # the caller's flags, untouched
    assert "re-executed itself" in seen["env"]["LV_SLAM_BENCH_LAUNCHER"]
    assert cmd == bench.launch_command(8, sys.argv[1:], cmd[cmd.index("--master-port") + 1])
    with pytest.raises(SystemExit) as e:                                              # 4 devices for 8 ranks: refuse, non-zero exit
        bench.self_launch(a, sys.argv[1:], visible_devices=4, execve=fake_execve)
    assert e.value.code not in (0, None) and "only 4 GPU(s) visible" in str(e.value.code)
    # a world size that contradicts --gpus is refused too (main reads WORLD_SIZE before anything else)
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=1" in str(e.value.code)


def test_hsa_ipc_mode_is_owned_by_bench_py():
    """RCCL's intra-node transport needs the dmabuf IPC mode on these hosts; bench.py sets the variable itself, before torch starts HSA."""
    import bench  # noqa: F401
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('setdefault("HSA_ENABLE_IPC_MODE_LEGACY"') < src.index("\nimport torch\n")


def test_kitti_loader_round_trip(tmp_path):
    """--kitti-dir's loader on a two-frame fake sequence: N x 4 f32 records, x,y,z kept bit for bit, ragged clouds packed into the
    engine's [pair][3][pitch] rows with their counts (scripts/lidar_odom_kitti.sh:6 is the reference's use of the same files)."""
    import numpy as np
    import pytest
    from lv_slam_amd import kitti
    vel = tmp_path / "sequences" / "04" / "velodyne"
    vel.mkdir(parents=True)
    rng = np.random.default_rng(3)
    a, b = rng.normal(size=(1000, 3)).astype(np.float32), rng.normal(size=(777, 3)).astype(np.float32)
    kitti.write_frame(str(vel / "000000.bin"), a, 0.25)
    kitti.write_frame(str(vel / "000001.bin"), b)
    (vel / "notes.txt").write_text("not a scan")
    files = kitti.list_frames(str(vel))
    assert [os.path.basename(f) for f in files] == ["000000.bin", "000001.bin"]
    assert kitti.list_frames(str(vel.parent)) == files                                # the sequence directory works too
    assert os.path.getsize(files[0]) == 1000 * 16
    assert np.array_equal(kitti.load_frame(files[0]), a) and np.array_equal(kitti.load_frame(files[1]), b)
    T, S, tc, sc, pitch = kitti.pack_soa([(a, b)], "cpu")
    assert (tc, sc, pitch) == ([1000], [777], 1024) and T.shape == (1, 3, 1024)
    assert np.array_equal(T[0, :, :1000].numpy().T, a) and np.array_equal(S[0, :, :777].numpy().T, b)
    assert float(S[0, :, 777:].abs().sum()) == 0.0                                    # padding is zero
    (vel / "000002.bin").write_bytes(b"\0" * 20)                                      # 1.25 records: refuse, do not truncate
    with pytest.raises(ValueError):
        kitti.load_frame(str(vel / "000002.bin"))
    with pytest.raises(FileNotFoundError):
        kitti.list_frames(str(tmp_path / "nowhere"))
