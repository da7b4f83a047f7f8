"""GPU (-m gpu): bench.py itself -- the N = 1 line's contract fields, and the N > 1 branch run as two ranks sharing cuda:0
(LV_SLAM_BENCH_BACKEND=gloo; the driver's own scaling runs use RCCL, one rank per GPU): round-robin sharding, pose records
packed on the device by the engine, the all-gather, and the two checks on it (pair ids form a permutation of the job, every
rank's own records come back bit-identical).  Independence of the pairs: scan_matching_odom_nodelet.cpp:240-250 is the only
coupling between frames in the reference, and the benchmark's fixed guess removes it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_bench(args, nproc=1, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_contract():
    d = run_bench(["--pairs", "6", "--azimuth", "256", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.5", "--config4-pairs", "9", "--seq-frames", "9"])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "registrations/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "reference_shaped_by_threads" in c
    assert c["value"] >= c["value_reference_shaped"] > 0 and "optimised port" in c["arrangement_of_value"]     # `value` = the fastest CPU arrangement
    p = d["parity"]
    assert p["pairs_checked"] >= 3 and p["iterations_equal"] == p["pairs_checked"] and p["max_dtrans_m"] < 1e-4 and p["max_drot_rad"] < 1e-5
    # the drop-in path's leg (host AoS clouds in, PCIe inclusive) is part of the default single-GPU line, and never `value`
    h = d["host_clouds"]
    assert d["value_host_clouds"] == max(h["registrations_per_s"], h["streamed"]["registrations_per_s"]) > 0 and h["registrations_per_s"] > 0
    assert ("streamed" in d["value_host_clouds_mode"]) == (h["streamed"]["registrations_per_s"] >= h["registrations_per_s"])
    assert h["bit_identical_to_device_resident_run"] is True and h["streamed"]["bit_identical_to_device_resident_run"] is True
    assert h["pairs_per_batch"] == 6 and h["record_bytes"] == 32
    # BASELINE config 4 rides along (here shrunk to 9 pairs): a strong-scaling block with its own parity sample
    c4 = d["config4"]
    assert c4["pairs_total"] == 9 and c4["scaling"] == "strong" and c4["value"] > 0 and c4["converged"] == 9
    assert c4["parity"]["pairs_checked"] >= 3 and c4["parity"]["iterations_equal"] == c4["parity"]["pairs_checked"]
    assert d["gather_ms_per_step"] is None and c4["gather_ms_per_step"] is None        # no process group in the plain N = 1 run
    # latency mode: the nodelet's per-frame loop on a drive, tracked on the device
    q = d["sequential"]
    assert d["value_sequential"] == q["frames_per_s"] > 0 and q["frames"] == 9 and q["aligns"] == 9 and q["converged"] == 8
    assert q["parity"]["frames_checked"] == 8 and q["parity"]["iterations_equal"] == 8 and q["parity"]["keyframe_decisions_equal"] == 8
    assert q["parity"]["max_dtrans_m"] < 1e-4 and q["parity"]["max_drot_rad"] < 1e-5 and q["host_round_trips_between_frames"] == 0
    # the process-group record: a plain single process, one rank, no backend
    assert d["world_size"] == 1 and d["process_group"]["backend"] is None and d["process_group"]["launcher"] == "plain process"
    assert d["process_group"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the secondary roofline figure (~450 flop per hit against the vector f32 peak)
    f = r["flops"]
    assert f["peak_tflops"] > 100 and abs(f["frac"] - f["achieved_tflops"] / f["peak_tflops"]) < 1e-3 and f["hits"] > 0
    # the other BASELINE configurations ride along in the same line (sizes follow --pairs / --azimuth: 6 x 16,384 and 2 x 32,768 here)
    oc = d["other_configs"]["configs"]
    assert set(oc) == {"ndt_pca_direct1", "config5_direct7", "config5_direct1"}
    for name, c in oc.items():
        assert c["value"] > 0 and c["ms_per_step"] > 0 and c["steps"] >= 5 and c["timed_s"] >= 0.2, name       # (the step count is fixed from the synchronous leg: the faster leg fills a little less than --other-seconds)
        rr = c["roofline"]
        assert rr["bound"] == "hbm" and 0 < rr["frac"] < 1.2 and rr["avg_launch_us"] > 0 and abs(rr["frac"] - rr["achieved"] / 8000.0) < 1e-3, name
        pp = c["parity"]
        assert pp["pairs_checked"] >= 2 and pp["iterations_equal"] == pp["pairs_checked"] and pp["max_dtrans_m"] < 1e-4 and pp["max_drot_rad"] < 1e-5, name
    assert "DIRECT1" in oc["config5_direct1"]["workload"] and "32768 pts" in oc["config5_direct7"]["workload"]
    d2 = run_bench(["--pairs", "4", "--azimuth", "256", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-host-clouds", "--config4-pairs", "0", "--seq-frames", "0",
                    "--no-other-configs"])
    assert "value_host_clouds" not in d2 and d2["config4"] is None and "value_sequential" not in d2 and d2["other_configs"] is None


def test_plain_gpus_2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` -- no torch.distributed.run in front, no WORLD_SIZE -- must come back as a TWO-rank job (it re-executes
    itself under the launcher), and says so: world_size as the process group reports it, the backend, who launched.  (gloo override: the
    two ranks share this box's one GPU; with the default RCCL backend the same command refuses when fewer than two devices are visible.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LV_SLAM_BENCH_BACKEND"] = "gloo"
    args = ["--gpus", "2", "--pairs", "3", "--azimuth", "256", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--config4-pairs", "5"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["process_group"]["backend"] == "gloo" and "re-executed itself" in d["process_group"]["launcher"]
    assert d["config"]["pairs_total"] == 6 and d["gather_check"]["pairs_gathered"] == 6 and d["gather_check"]["world_size"] == 2
    assert d["config4"]["gather_check"]["pairs_gathered"] == 5
    import torch
    if torch.cuda.device_count() < 2:                       # the default backend on a one-GPU box: refuse, never a one-rank run under that name
        env.pop("LV_SLAM_BENCH_BACKEND")
        bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert bad.returncode != 0 and "GPU(s) visible" in bad.stderr and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_eight_ranks_on_one_device_gloo():
    """The scaling run's largest shape, as far as one GPU allows: `bench.py --gpus 8` as EIGHT ranks (gloo; all on cuda:0), the weak-scaling job
    (2 pairs per rank, streamed) and the config-4-style fixed job of 13 pairs (uneven: five ranks own 2, three own 1 + a padding row)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LV_SLAM_BENCH_BACKEND"] = "gloo"
    args = ["--gpus", "8", "--pairs", "2", "--azimuth", "128", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--config4-pairs", "13"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["world_size"] == 8 and d["process_group"]["world_size"] == 8 and d["process_group"]["backend"] == "gloo"
    assert d["config"]["pairs_total"] == 16 and d["config"]["pairs_rank0"] == 2
    g = d["gather_check"]
    assert g["pairs_gathered"] == 16 and g["permutation_of_all_pair_ids"] and g["own_records_bit_identical_on_every_rank"] and g["world_size"] == 8
    c4 = d["config4"]
    assert c4["pairs_total"] == 13 and c4["pairs_rank0"] == 2 and c4["n_gpus"] == 8
    g4 = c4["gather_check"]
    assert g4["pairs_gathered"] == 13 and g4["records_per_rank"] == 2 and g4["permutation_of_all_pair_ids"] and g4["own_records_bit_identical_on_every_rank"]
    assert d["config"]["stream"]["bit_identical_to_synchronous"] is True and d["value_synchronous"] > 0
    # every rank reports how long its inputs took (the ranks of a node share the host's CPUs): eight values, none out of line
    g = d["config"]["input_generation_s_per_rank"]
    assert len(g) == 8 and d["config"]["input_generation_s"] == max(g) and max(g) < 20.0, g


def test_streamed_headline_carries_both_rates():
    """The N = 1 line: `value` is the streamed job (distinct batches back to back through mi355ndt_stream_*), `value_synchronous` the same steps one
    batch at a time; the streamed results are bit-identical to the synchronous ones, pairs really were handed from launch to launch."""
    d = run_bench(["--pairs", "40", "--azimuth", "512", "--steps", "6", "--warmup", "1", "--cpu-seconds", "0", "--no-host-clouds", "--config4-pairs", "0", "--seq-frames", "0",
                   "--no-other-configs"])
    st = d["config"]["stream"]
    assert st["n_batches"] == 4 and st["n_contexts"] == 4 and st["bit_identical_to_synchronous"] is True and st["batches_rerun"] == 0 and st["launches_that_gave_up"] == 0
    assert st["launches"] >= 6 and st["pairs_handed_over"] > 0
    assert d["value_streamed"] > 0 and d["value_synchronous"] > 0 and d["value"] == d["value_streamed"] and d["value_mode"] == "streamed"
    assert d["config"]["mode"].startswith("streamed")
    assert d["roofline_streamed"]["launches"] >= 6 and d["roofline_synchronous"]["launches"] == 6
    d0 = run_bench(["--pairs", "8", "--azimuth", "256", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-host-clouds", "--config4-pairs", "0", "--seq-frames", "0",
                    "--no-other-configs", "--no-stream"])
    assert d0["config"]["stream"] is None and d0["value"] == d0["value_synchronous"] and d0["value_streamed"] is None and d0["config"]["mode"].startswith("synchronous")


def test_kitti_directory_as_input(tmp_path):
    """--kitti-dir: a KITTI-shaped sequence (here five synthetic scans written as velodyne/%06d.bin, one of them cut short so the
    clouds are ragged) -> consecutive frame pairs -> the same timed step, `data` = "kitti", parity against the oracle on the same clouds."""
    import numpy as np
    from lv_slam_amd import kitti, synth
    vel = tmp_path / "sequences" / "04" / "velodyne"
    vel.mkdir(parents=True)
    scans, _ = synth.make_sequence(5, 256)
    for k, sc in enumerate(scans):
        xyz = sc.numpy()
        kitti.write_frame(str(vel / f"{k:06d}.bin"), xyz[: len(xyz) - 1000 * (k == 2)])
    d = run_bench(["--kitti-dir", str(vel), "--pairs", "4", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2", "--config4-pairs", "0", "--seq-frames", "0"])
    assert d["data"] == "kitti" and "KITTI seq 04" in d["config"]["workload"] and d["config"]["pairs_total"] == 4
    assert d["config"]["points_per_cloud"] == 16384 and 16384 - 250 - 1 < d["config"]["mean_points_per_source"] < 16384
    p = d["parity"]
    assert p["pairs_checked"] >= 3 and p["iterations_equal"] == p["pairs_checked"] and p["max_dtrans_m"] < 1e-4 and p["max_drot_rad"] < 1e-5
    assert d["other_configs"] is None and "value_host_clouds" not in d
    dp = run_bench(["--kitti-dir", str(vel), "--kitti-prefilter", "--pairs", "3", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--config4-pairs", "0", "--seq-frames", "0"])
    assert dp["data"] == "kitti" and 0 < dp["config"]["mean_points_per_source"] < 16384 and "prefilter" in dp["config"]["inputs"]


@pytest.mark.parametrize("extra,total", [(["--pairs", "5"], 10), (["--total-pairs", "11"], 11)])
def test_two_ranks_on_one_gpu_gather(extra, total):
    """weak scaling (5 pairs per rank) and the config-4-literal mode with uneven shards (11 pairs: 6 + 5)."""
    d = run_bench(["--gpus", "2", "--azimuth", "256", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--config4-pairs", "13"] + extra, nproc=2,
                  env_extra={"LV_SLAM_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["pairs_total"] == total
    assert d["scaling"] == ("strong" if "--total-pairs" in extra else "weak")
    g = d["gather_check"]
    assert g["pairs_gathered"] == total and g["permutation_of_all_pair_ids"] is True and g["own_records_bit_identical_on_every_rank"] is True
    assert g["record_bytes"] == 96 and g["packed_on_device"] is True
    assert d["config"]["converged"] == d["config"]["pairs_rank0"]
    assert d["gather_ms_per_step"] is not None and d["gather_ms_per_step"] >= 0       # the gather time listed separately
    c4 = d["config4"]
    if "--total-pairs" in extra:
        assert c4 is None                              # the line itself is the fixed job
    else:                                              # the one line a driver that only passes --gpus N gets: weak value + config-4 block
        assert c4["pairs_total"] == 13 and c4["pairs_rank0"] == 7 and c4["scaling"] == "strong" and c4["n_gpus"] == 2 and c4["value"] > 0
        g4 = c4["gather_check"]
        assert g4["pairs_gathered"] == 13 and g4["permutation_of_all_pair_ids"] is True and g4["own_records_bit_identical_on_every_rank"] is True
        assert c4["gather_ms_per_step"] is not None and c4["gather_ms_per_step"] >= 0 and c4["converged"] == 7


def test_single_rank_through_rccl():
    """The N > 1 code path with the backend the driver's scaling runs use (nccl = RCCL), as far as one GPU can take it: a single
    rank made to go through the process group, the records packed on the device and all_gather_into_tensor on device memory."""
    d = run_bench(["--pairs", "7", "--azimuth", "256", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-host-clouds", "--seq-frames", "0"],
                  env_extra={"LV_SLAM_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(free_port())})
    g = d["gather_check"]
    assert g["backend"] == "nccl" and g["host_hop"] is False and g["packed_on_device"] is True
    assert g["pairs_gathered"] == 7 and g["permutation_of_all_pair_ids"] is True and g["own_records_bit_identical_on_every_rank"] is True
    assert d["n_gpus"] == 1 and d["config"]["converged"] == 7
    assert d["gather_ms_per_step"] is not None and d["gather_ms_per_step"] > 0 and "HIP events" in g["timed_with"]


def test_config4_job_on_one_gpu_parity_over_the_index_range_and_rccl_gather():
    """BASELINE config 4's job -- 4,541 pairs of 65,536 points, ndt_omp, 1 m, DIRECT7 -- as ONE batch on one GPU: the parity leg samples the
    WHOLE index range (every ~16th pair incl. 4540, not the first 271), and the single rank is made to go through RCCL so that the
    gather of 4,541 device-packed 96-byte records is checked too.  Independence of the pairs: scan_matching_odom_nodelet.cpp:240-250."""
    d = run_bench(["--total-pairs", "4541", "--steps", "3", "--warmup", "1", "--cpu-seconds", "10", "--no-host-clouds", "--seq-frames", "0"],
                  env_extra={"LV_SLAM_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(free_port())})
    assert d["scaling"] == "strong" and d["config"]["pairs_total"] == 4541 and d["config"]["pairs_rank0"] == 4541 and d["config"]["converged"] == 4541
    p = d["parity"]
    assert p["pairs_checked"] >= 285 and p["iterations_equal"] == p["pairs_checked"] and p["converged_flags_equal"] == p["pairs_checked"]
    assert p["max_dtrans_m"] < 1e-4 and p["max_drot_rad"] < 1e-5 and "..4540" in p["sample"]
    g = d["gather_check"]
    assert g["backend"] == "nccl" and g["pairs_gathered"] == 4541 and g["permutation_of_all_pair_ids"] is True
    assert g["own_records_bit_identical_on_every_rank"] is True and g["records_per_rank"] == 4541 and g["host_hop"] is False
    assert d["gather_ms_per_step"] > 0


@pytest.mark.parametrize("order", [0, 1])
def test_default_workload_parity_leg_covers_the_whole_batch(order):
    """BASELINE config 3 as bench.py runs it by default (271 pairs x 65,536 pts, ndt_omp, 1 m, DIRECT7): the line's parity leg
    checks every pair of the batch against the oracle -- same iteration counts, SE(3) inside the north-star tolerance -- and the
    roofline block is internally consistent.  Both evaluation orders of the three-term f32 sums (ndt_omp_impl2.hpp:581, 594-613):
    the canonical one, and the lane pairing of Eigen 3.3's SSE predux against the oracle's matching variant."""
    d = run_bench(["--steps", "5", "--warmup", "2", "--cpu-seconds", "30", "--config4-pairs", "0", "--seq-frames", "0", "--no-other-configs", "--no-host-clouds",
                   "--f32-sum-order", str(order)])
    p = d["parity"]
    assert p["f32_sum_order"] == order and d["config"]["f32_sum_order"] == order
    assert p["pairs_checked"] == 271 and p["iterations_equal"] == 271 and p["max_dtrans_m"] < 1e-4 and p["max_drot_rad"] < 1e-5
    assert d["config"]["converged"] == 271 and d["config"]["pairs_total"] == 271
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3 and r["hits_per_point"] > 3.5
    assert abs(r["alg_bytes_per_launch"] * r["launches"] / (r["sweep_ms_per_step"] * 1e-3 * d["steps"]) / 1e9 - r["achieved"]) < 0.02 * r["achieved"]
    c = d["cpu_baseline"]
    assert set(c["reference_shaped_by_threads"]) == set(c["optimised_port_by_threads"]) and c["value"] > 0


def test_prefiltered_row():
    """bench.py --prefiltered: the reference-conditioned row (raw scans -> device prefilter as launch/dlo_kitti.launch:30-36 sets it ->
    target / source -> align), with the oracle's prefilter + align as the checker."""
    d = run_bench(["--prefiltered", "--azimuth", "512", "--pairs", "4", "--variant", "pca", "--mode", "direct1", "--cpu-seconds", "2"])
    assert d["value"] > 0 and d["registrations_per_s_without_prefilter_time"] > d["value"] and d["config"]["converged"] == 4
    c = d["config"]
    assert c["raw_points_per_cloud"] == 32768 and 0 < c["filtered_points_per_target_mean"] < 32768 and c["searchable_leaves_per_target_mean"] > 50
    p = d["parity"]
    assert p["pairs_checked"] >= 2 and p["iterations_equal"] == p["pairs_checked"] and p["filtered_point_counts_equal"] == p["pairs_checked"]
    assert p["max_dtrans_m"] < 1e-4 and p["max_drot_rad"] < 1e-5
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0


def test_tolerance_mode_block_rides_in_the_line():
    """The default line carries the same jobs under MI355NDT_OPT_ARITH = 1 beside the exact `value` (which stays the headline): its own rates (streamed and
    synchronous), every pair's result against the exact arithmetic's, a bounded oracle sample -- for the headline, the config-4 job and every other configuration."""
    d = run_bench(["--pairs", "24", "--azimuth", "512", "--steps", "4", "--warmup", "1", "--cpu-seconds", "4", "--no-host-clouds", "--config4-pairs", "29", "--seq-frames", "0",
                   "--other-seconds", "0.05"])
    assert d["config"]["arith"] == 0 and d["dtype"] == "f32 terms, f64 accumulation" and d["value"] == d["value_streamed"]
    t = d["tolerance_mode"]
    assert d["value_tolerance_mode"] == t["value_tolerance_mode"] == t["value_tolerance_mode_streamed"] > 0 and t["value_tolerance_mode_synchronous"] > 0
    assert t["stream"]["bit_identical_to_synchronous"] is True
    v = t["vs_exact_arithmetic"]
    assert v["pairs"] == 24 and v["iteration_flips"] == 0 and v["pairs_beyond_tolerance"] == 0 and 0 < v["max_dtrans_m"] < 1e-4
    p = t["parity_vs_oracle"]
    assert p["arith"] == 1 and p["pairs_checked"] >= 3 and p["iterations_equal"] == p["pairs_checked"] and p["pairs_beyond_tolerance"] == 0
    c4 = d["config4"]["tolerance_mode"]
    assert c4["vs_exact_arithmetic"]["pairs"] == 29 and c4["vs_exact_arithmetic"]["pairs_beyond_tolerance"] == 0 and d["config4"]["value_tolerance_mode"] > 0
    for name, c in d["other_configs"]["configs"].items():
        assert c["value_mode"] == "streamed" and c["value"] == c["value_streamed"], name
        assert c["tolerance_mode"]["vs_exact_arithmetic"]["iteration_flips"] == 0 and c["tolerance_mode"]["vs_exact_arithmetic"]["pairs_beyond_tolerance"] == 0, name
    # the whole line in the tolerance arithmetic, on request, says so
    d1 = run_bench(["--arith", "1", "--pairs", "8", "--azimuth", "256", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2", "--no-host-clouds", "--config4-pairs", "0", "--seq-frames", "0",
                    "--no-other-configs"])
    assert d1["config"]["arith"] == 1 and "tolerance arithmetic" in d1["dtype"] and d1["tolerance_mode"] is None
    assert d1["parity"]["arith"] == 1 and d1["parity"]["pairs_beyond_tolerance"] == 0 and d1["parity"]["iterations_equal"] == d1["parity"]["pairs_checked"]
