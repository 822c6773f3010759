"""The A/B switches select alternative kernels for the same arithmetic: the matrix-core matcher and vote loop must reproduce the
vector-ALU ones bit for bit; the solver's scheduling / preconditioner variants must land on the same poses within the parity bar."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from uzliti_slam_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(**env):
    # the switches only exist in the diagnostic build of the library (csrc/Makefile target `diag`, -DUZL_DIAG)
    diag = os.path.join(os.path.dirname(HERE), "uzliti_slam_amd", "libuzl_mi355x_diag.so")
    assert os.path.exists(diag), "build the diagnostic library: make -C uzliti_slam_amd/csrc diag"
    e = dict(os.environ, UZL_LIB=diag, **env)
    out = subprocess.run([sys.executable, os.path.join(HERE, "_ab_worker.py")], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_alternative_kernel_paths_agree():
    ref = _run()
    # the shipped library ignores the switches: same digests as the diagnostic build without any switch set
    e = dict(os.environ, UZL_KNN2_VALU="1", UZL_VOTE_VALU="1", UZL_ML_NO_COMP4="1", UZL_ML_ADDITIVE="1")
    e.pop("UZL_LIB", None)
    out = subprocess.run([sys.executable, os.path.join(HERE, "_ab_worker.py")], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    shipped = json.loads(out.stdout.strip().splitlines()[-1])
    assert shipped["match"] == ref["match"] and shipped["large"]["pcg"] == ref["large"]["pcg"] and shipped["small"]["poses"] == ref["small"]["poses"]
    valu = _run(UZL_KNN2_VALU="1", UZL_VOTE_VALU="1")
    assert valu["match"] == ref["match"], "matrix-core matcher / votes differ from the vector-ALU kernels"
    alt = _run(UZL_ML_SYNC_REBUILD="1", UZL_ML_NO_COMP4="1")
    for k in ("small", "large"):
        assert ref[k]["status"] == alt[k]["status"] == 0
        a = np.array(ref[k]["poses"]).reshape(-1, 3, 4); b = np.array(alt[k]["poses"]).reshape(-1, 3, 4)
        dt, dr = synth.pose_errors(a, b)
        assert dt < 1e-3 and dr < 1e-4, (k, dt, dr)
        assert abs(ref[k]["chi2"] - alt[k]["chi2"]) <= 1e-5 * abs(alt[k]["chi2"])
    # the dense level-2 operator is what makes the large-graph path converge in half the iterations
    assert ref["large"]["pcg"] < 0.75 * alt["large"]["pcg"]
