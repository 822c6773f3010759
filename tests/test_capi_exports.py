"""CPU tests (no GPU): the product C-ABI library loads and exports every symbol include/uzl_mi355x.h
declares; without a GPU it fails loudly instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "uzl_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(uzl_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_every_declared_symbol_is_exported(capi):
    L = ctypes.CDLL(capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_nothing_but_the_header_is_exported(capi):
    """the reference-side binding sees uzl_* and nothing else - no kernel stubs, no internals, no test hooks."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(line.split()[-1] for line in out.splitlines() if line.strip()))
    extra = [n for n in exported if n not in set(_declared())]
    assert not extra, extra[:20]
    assert not [n for n in exported if "debug" in n]


def test_struct_layouts_match_header(capi):
    assert ctypes.sizeof(capi.EdgeResult) == capi.EDGE_RESULT_DTYPE.itemsize == 432
    assert ctypes.sizeof(capi.PairJob) == capi.PAIR_JOB_DTYPE.itemsize == 24
    assert capi.NODE_DTYPE.itemsize == 104 and capi.EDGE_DTYPE.itemsize == 608
    assert ctypes.sizeof(capi.MatchCfg) == 56 and ctypes.sizeof(capi.PgoCfg) == 64


def test_defaults_mirror_the_cfg_files(capi):
    c = capi.MatchCfg(); capi.lib().uzl_match_cfg_default(ctypes.byref(c))
    # transformation_estimation/cfg/FeatureLinkEstimation.cfg:9-13
    assert (c.ransac_threshold, c.link_covariance, c.ransac_iteration, c.ransac_break_percentage, c.use_epnp) == (0.2, 0.01, 100, 0.6, 1)
    p = capi.PgoCfg(); capi.lib().uzl_pgo_cfg_default(ctypes.byref(p))
    # graph_optimization/cfg/GraphOptimizer.cfg:10-12
    assert (p.iterations, p.use_odometry_parameters, p.optimize_xy_only, p.huber_delta) == (20, 0, 0, 1.0)
    assert capi.lib().uzl_abi_version() == 3
    assert capi.lib().uzl_status_string(-1) == b"bad argument"


def test_no_gpu_means_loud_failure(capi):
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    assert capi.device_count() == capi.UZL_ERR_NO_DEVICE
    with pytest.raises(capi.UzlError) as e:
        capi.Match()
    assert e.value.status == capi.UZL_ERR_NO_DEVICE
    with pytest.raises(capi.UzlError) as e:
        capi.Pgo()
    assert e.value.status == capi.UZL_ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under uzliti_slam_amd/ may import, include or link it."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "uzliti_slam_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"(import\s+oracle|from\s+oracle|uzl_oracle\.h|libuzl_oracle|uzlo_)", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
