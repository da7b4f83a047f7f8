"""CPU: the randomised differential tester's scene / parameter generators and its oracle side run (dry run, no GPU), and its
summary format -- so that tools/fuzz_parity.py stays runnable between the GPU sessions that use it."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_tool_dry_run_and_summary(tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--cases", "14", "--seed", "5", "--oracle-only", "--aux", "0.5"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "fuzz: 14 cases, 0 failures" in out.stdout and "0 errors" in out.stdout, out.stdout[-2000:]
    # the committed summaries have the fields the docs quote
    for name in ("r03_fuzz.json", "r03_fuzz_aux.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d["cases"] > 1000 and d["errors"] == 0 and d["failures_not_explained_by_the_oracles_own_order_sensitivity"] == 0
        assert all(f["variant"] == 1 and f["neighbor_mode"] == 1 for f in d["failing_aligns"])        # ndt_pca + DIRECT26 only
        assert d["totals"]["voxel_checks"] == d["totals"]["sweep_checks"] > 500
