"""Build-container-only: the committed fixtures under tests/golden/ ARE what the reference's own code produces.  Both
generators (gen_mpm_golden.py: the reference's @wp.kernel bodies of modules/nclaw/sim/mpm.py:321-498, interface.py:126-147,
warp/svd.py:61-96, d3gs/utils/simulation_utils.py:25-48 executed on the scalar `wp` stand-in; gen_material_golden.py: the
reference's material / camera / scheduler classes imported from /root/reference) are re-run into a temporary directory and
every array of every .npz must equal the committed one exactly.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

REF = Path("/root/reference")
GOLD = Path(__file__).resolve().parent / "golden"

pytestmark = pytest.mark.skipif(not (REF / "modules" / "nclaw" / "sim" / "mpm.py").exists(), reason="needs the reference checkout (build container only)")


def test_generators_reproduce_the_committed_fixtures(tmp_path):
    (tmp_path / "data").mkdir()
    env = dict(os.environ, NEUMA_GOLDEN_OUT=str(tmp_path), OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    jobs = [[sys.executable, str(GOLD / "gen_mpm_golden.py"), "steps"],
            [sys.executable, str(GOLD / "gen_mpm_golden.py"), "grads"],
            [sys.executable, str(GOLD / "gen_mpm_golden.py"), "rollout", "init", "svd", "cov"],
            [sys.executable, str(GOLD / "gen_material_golden.py")]]
    procs = [subprocess.Popen(c, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for c in jobs]
    for c, p in zip(jobs, procs):
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, f"{c[1:]}: rc {p.returncode}\n{out[-2000:]}"
    made = sorted(f.name for f in tmp_path.glob("*.npz"))
    committed = sorted(f.name for f in GOLD.glob("*.npz"))
    assert made == committed, (set(made) ^ set(committed))
    for name in committed:
        a, b = np.load(GOLD / name, allow_pickle=False), np.load(tmp_path / name, allow_pickle=False)
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (name, k)
            assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (name, k)
    # the product's copy of the three shipped checkpoints is the same data
    a, b = np.load(GOLD / "base_models.npz"), np.load(tmp_path / "data" / "base_models.npz")
    assert all(np.array_equal(a[k], b[k]) for k in a.files)
    shipped = np.load(GOLD.parent.parent / "neuma_amd" / "data" / "base_models.npz")
    assert sorted(shipped.files) == sorted(a.files) and all(np.array_equal(a[k], shipped[k]) for k in a.files)
