"""CPU: the C-ABI library loads, exports every symbol include/mi355_ndt.h declares, and refuses to run
without a GPU (no compute calls here)."""
import os
import re
import ctypes as C
import pytest

from conftest import ROOT
from lv_slam_amd import ndt


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(ndt.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return ndt.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mi355_ndt.h")).read()
    declared = sorted(set(re.findall(r"\b(mi355ndt_[a-z_A-Z0-9]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    for s in declared:
        assert hasattr(lib, s), f"{s} declared in mi355_ndt.h but not exported"
    assert sorted(ndt.SYMBOLS) == declared


def test_struct_layouts_match_header(lib, tmp_path):
    # compile the header with gcc and compare sizeof/offsetof with the ctypes mirrors
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "mi355_ndt.h"
int main(void) {
  printf("%zu %zu %zu %zu ", sizeof(mi355ndt_params), sizeof(mi355ndt_result), sizeof(mi355ndt_voxel), sizeof(mi355ndt_profile));
  printf("%zu %zu %zu %zu ", offsetof(mi355ndt_params, min_covar_eigvalue_mult), offsetof(mi355ndt_result, hits_last),
         offsetof(mi355ndt_voxel, weight), offsetof(mi355ndt_profile, update_launches));
  printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(mi355ndt_seq_params), sizeof(mi355ndt_seq_frame), sizeof(mi355ndt_seq_stats),
         offsetof(mi355ndt_seq_frame, tf_s2k_colmajor), offsetof(mi355ndt_seq_frame, key_id), offsetof(mi355ndt_seq_stats, update_launches));
  return 0;
}''')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    exp = [C.sizeof(ndt.Params), C.sizeof(ndt.Result), C.sizeof(ndt.Voxel), C.sizeof(ndt.Profile),
           ndt.Params.min_covar_eigvalue_mult.offset, ndt.Result.hits_last.offset, ndt.Voxel.weight.offset,
           ndt.Profile.update_launches.offset,
           C.sizeof(ndt.SeqParams), C.sizeof(ndt.SeqFrame), C.sizeof(ndt.SeqStats), ndt.SeqFrame.tf_s2k_colmajor.offset, ndt.SeqFrame.key_id.offset,
           ndt.SeqStats.update_launches.offset]
    assert got == exp


def test_default_params_are_reference_ctor_defaults(lib):
    p = ndt.default_params()
    # ndt_omp_impl2.hpp:53-83; voxel_grid_covariance_omp.h:204-205
    assert (p.resolution, p.step_size, p.outlier_ratio, p.trans_epsilon, p.max_iterations) == (1.0, 0.1, 0.55, 0.1, 35)
    assert p.neighbor_mode == ndt.DIRECT7 and p.variant == ndt.VARIANT_OMP
    assert p.min_points_per_voxel == 6 and p.min_covar_eigvalue_mult == 0.01


def test_no_cpu_fallback(lib):
    if lib.mi355ndt_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ndt.NDTError) as e:
        ndt.Engine()
    assert e.value.code == -5
    with pytest.raises(ndt.NDTError):
        ndt.NormalDistributionsTransform()
    # NULL handle is rejected, not dereferenced
    assert lib.mi355ndt_align(None, None, None) == -1
    assert lib.mi355ndt_destroy(None) == -1


def test_product_never_imports_oracle():
    # the product path must not route through the oracle (only tests/, smoke(), bench cpu_baseline may)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lv_slam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} mentions the oracle"
    assert "oracle" not in open(os.path.join(ROOT, "include", "mi355_ndt.h")).read()


def test_mirror_has_reference_method_names():
    """the host mirror answers to the reference's own method names (ndt_omp.h:109-203, ndt_pca.h:129-133, pcl::Registration)"""
    names = ["setNumThreads", "setResolution", "getResolution", "setStepSize", "getStepSize", "setOulierRatio", "getOulierRatio",
             "setNeighborhoodSearchMethod", "getTransformationProbability", "getFinalNumIteration", "getTargetCells",
             "setInputTarget", "setInputSource", "align", "getFinalTransformation", "hasConverged", "getFitnessScore",
             "setTransformationEpsilon", "setMaximumIterations"]
    for n in names:
        assert callable(getattr(ndt.NormalDistributionsTransform, n, None)), n
    # enum order of ndt_omp.h:51-56
    assert (ndt.KDTREE, ndt.DIRECT26, ndt.DIRECT7, ndt.DIRECT1) == (0, 1, 2, 3)



def test_option_and_warning_constants_match_header(lib, tmp_path):
    """the engine options and the one positive status value, as the header spells them, are what the python binding uses; the profile's cloud counters sit
    where the header puts them"""
    import subprocess
    src = tmp_path / "opt.c"
    src.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "mi355_ndt.h"
int main(void) {
  printf("%d %d %d %d %d %d %d %d %d ", MI355NDT_OPT_F32_SUM_ORDER, MI355NDT_OPT_ASYNC_ALIGN, MI355NDT_OPT_DEBUG_ASYNC_ABORT, MI355NDT_OPT_STREAM_THRESHOLD,
         MI355NDT_OPT_STREAM_RESERVE, MI355NDT_OPT_DEBUG_ASYNC_RINGS, MI355NDT_OPT_ARITH, MI355NDT_WARN_TOLERANCE_ARITH, MI355NDT_TOLERANCE_MIN_HITS);
  printf("%zu %zu %zu %zu %zu\\n", offsetof(mi355ndt_profile, cloud_uploads), offsetof(mi355ndt_profile, cloud_transfers), offsetof(mi355ndt_profile, cloud_promotions),
         offsetof(mi355ndt_profile, stream_launch_slots), sizeof(mi355ndt_profile));
  return 0;
}''')
    exe = tmp_path / "opt"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [ndt.OPT_F32_SUM_ORDER, ndt.OPT_ASYNC_ALIGN, ndt.OPT_DEBUG_ASYNC_ABORT, ndt.OPT_STREAM_THRESHOLD, ndt.OPT_STREAM_RESERVE, ndt.OPT_DEBUG_ASYNC_RINGS,
                   ndt.OPT_ARITH, ndt.WARN_TOLERANCE_ARITH, 4096,
                   ndt.Profile.cloud_uploads.offset, ndt.Profile.cloud_transfers.offset, ndt.Profile.cloud_promotions.offset,
                   ndt.Profile.stream_launch_slots.offset, C.sizeof(ndt.Profile)]
    assert ndt.WARN_TOLERANCE_ARITH > 0      # a caveat on a result, never an error code
