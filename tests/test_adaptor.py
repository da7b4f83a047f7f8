"""include/mi355_ndt_pcl.hpp -- the pcl::Registration adaptor a ROS/PCL host compiles -- built against
tests/pcl_stub (a stand-in for the few PCL/Eigen declarations the adaptor touches: there is no PCL in this image).

CPU: the adaptor compiles warning-free as C++14 and links against libmi355ndt.so (every C-ABI symbol it uses resolves).
GPU: the compiled adaptor, driven the way scan_matching_odom_nodelet.cpp:109-119,197,220-226 drives its registration
object, returns what the python mirror and the oracle return, incl. transformation_ / previous_transformation_
(ndt_omp_impl2.hpp:134,163) and PCL's align() post-conditions on the output cloud."""
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "pcl_stub")
EXE = os.path.join(STUB, "adaptor_main")


def build_adaptor():
    import __graft_entry__ as entry
    entry.build()
    return entry.build_adaptor_test()


def test_adaptor_compiles_and_links_against_pcl_stub():
    exe = build_adaptor()
    assert os.path.exists(exe)
    # the ABI symbols the adaptor needs are undefined in the binary and exported by the library
    und = subprocess.check_output(["nm", "-u", exe], text=True)
    for sym in ["mi355ndt_create", "mi355ndt_set_target", "mi355ndt_set_source", "mi355ndt_align", "mi355ndt_get_aligned",
                "mi355ndt_get_incremental", "mi355ndt_set_params", "mi355ndt_get_fitness_score", "mi355ndt_get_voxels", "mi355ndt_set_latency_mode",
                "mi355ndt_calculate_score", "mi355ndt_convert_transform", "mi355ndt_promote_source_to_target", "mi355ndt_profile_get"]:
        assert sym in und, sym


@pytest.mark.gpu
@pytest.mark.parametrize("variant,mode,res,lat", [(0, 2, 1.0, 0), (1, 3, 1.0, 0), (1, 3, 1.0, 1)])      # ndt_omp/DIRECT7, the nodelet's ndt_pca/DIRECT1, and the latter in latency mode
def test_adaptor_end_to_end(tmp_path, variant, mode, res, lat):
    from conftest import se3_err
    from lv_slam_amd import ndt, synth
    from oracle import oracle_py as O
    exe = build_adaptor()
    tgt, src, dT = synth.make_pair(3, 256, n_beams=32)
    tgt, src = tgt.numpy().astype(np.float32), src.numpy().astype(np.float32)
    tgt.tofile(tmp_path / "t.f32")
    src.tofile(tmp_path / "s.f32")
    out = subprocess.check_output([exe, str(tmp_path / "t.f32"), str(tmp_path / "s.f32"), str(len(tgt)), str(len(src)),
                                   str(variant), str(mode), str(res), str(lat)], text=True, timeout=300).strip().splitlines()
    v = out[0].split()
    F = np.array(v[:16], np.float32).reshape(4, 4).T
    L = np.array(v[16:32], np.float32).reshape(4, 4).T
    iters, conv, tp, vis, fit = int(v[32]), int(v[33]), float(v[34]), int(v[35]), float(v[36])
    kw = dict(resolution=res, trans_epsilon=0.01, max_iterations=64, neighbor_mode=mode, variant=variant)
    G = synth.default_guess()
    eng = ndt.Engine(ndt.default_params(**kw))
    eng.set_latency_mode(bool(lat))
    eng.set_target(tgt)
    eng.set_source(src)
    r = eng.align(G)
    assert np.array_equal(F, r["final"]) and iters == r["iterations"] and conv == int(r["converged"])
    assert tp == r["trans_probability"]
    inc, prev = eng.get_incremental()
    assert np.array_equal(L, inc)
    ro = O.align(O.Grid(tgt, O.default_params(**kw)), src, G)
    assert iters == ro["iterations"]
    dt, dr = se3_err(ro["final"], F)
    assert dt < 1e-4 and dr < 1e-5
    # transformation_ / previous_transformation_: the oracle's float(exp(delta_p)) of the last two steps
    for got, want in ((inc, ro["transformation"]), (prev, ro["previous_transformation"])):
        dt, dr = se3_err(want, got)
        assert dt < 1e-6 and dr < 1e-6, (dt, dr)
    assert vis == 1
    assert abs(fit - eng.fitness_score(4.0)[0]) <= 1e-12 * max(1.0, abs(fit))
    # align() post-conditions: x,y,z moved by the final pose (f32), data[3] = 1, the other fields of the record kept
    a = eng.get_aligned()
    for i in range(4):
        p = np.array(out[1 + i].split(), np.float64)
        assert np.array_equal(p[:3].astype(np.float32), a[i]) and p[3] == 1.0 and p[4] == float(i)
    assert int(out[5]) == eng.get_grid()[3]
    # calculateScore(output cloud) and the static convertTransform through the adaptor == through the python mirror
    w = out[6].split()
    full = ndt.NormalDistributionsTransform(variant=variant)
    assert float(w[0]) == eng.calculate_score(a)
    assert np.array_equal(np.array(w[1:17], np.float32).reshape(4, 4).T, full.convertTransform([1.0, 2.0, 3.0, 0.1, 0.2, 0.3])) and int(w[17]) == 1
    full.engine.close()


@pytest.mark.gpu
def test_adaptor_sends_every_frame_over_pcie_once(tmp_path):
    """The nodelet's call sequence (scan_matching_odom_nodelet.cpp:192-261) through the adaptor: frame 0 is uploaded as the first keyframe, every later frame
    ONCE as the source (setInputSource hands it over; neither align() -- frame 1 runs two -- nor the keyframe switch `key = filtered; setInputTarget(key)`
    sends it again: the switch is a device-to-device promotion).  Poses equal those of the plain C-ABI sequence set_target / set_source / align."""
    from lv_slam_amd import ndt, synth
    exe = build_adaptor()
    nf, naz = 8, 256
    scans, _ = synth.make_sequence(nf, naz, n_beams=32)
    frames = np.stack([sc.numpy().astype(np.float32) for sc in scans])
    n = frames.shape[1]
    frames.tofile(tmp_path / "frames.f32")
    out = subprocess.check_output([exe, "seq", str(tmp_path / "frames.f32"), str(nf), str(n)], text=True, timeout=300).strip().splitlines()
    uploads, promotions, nbytes = (int(x) for x in out[0].split())
    assert uploads == nf, out[0]                          # one per frame: the first as target, the others as source
    assert promotions == (nf - 1) // 3 and nbytes == nf * n * 12
    # the same sequence through the C-ABI with an upload for everything
    eng = ndt.Engine(ndt.default_params(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT1, variant=1))
    eng.set_target(frames[0])
    G = synth.default_guess()
    for k in range(1, nf):
        eng.set_source(frames[k])
        r = eng.align(G)
        if k == 1:
            r = eng.align(r["final"])
        assert np.array_equal(np.array(out[k].split(), np.float32).reshape(4, 4).T, r["final"]), k
        if k % 3 == 0:
            eng.set_target(frames[k])
    assert eng.profile_get()["cloud_uploads"] == nf + (nf - 1) // 3      # (this sequence sent every keyframe twice)
    # ... and the promotion on its own: the grid it leaves is the grid set_target builds from the same cloud
    v_up = eng.get_voxels(0)
    eng.set_source(frames[6])
    eng.promote_source_to_target()
    v_pr = eng.get_voxels(0)
    assert v_up.tobytes() == v_pr.tobytes() and eng.profile_get()["cloud_promotions"] == 1
    eng.close()
