"""Admission boundaries of the dense level-2 operator's PCG variant (csrc/pgo_types.hpp: kMlLdsLimit, ml_comp4_lds; pgo_ml_kernels.hip:
ml_comp4_fits) - ONE statement of the LDS budget that build_ml, k_ml_cg, ml_cg_variant and ml_fits_lds read (round 5 had four copies).
Host arithmetic of the diagnostic build: runs without a device."""
import ctypes as C
import os

import pytest

from uzliti_slam_amd import capi


def _adm(nb, n2):
    if not os.path.exists(capi.DIAG_LIB_PATH):
        pytest.skip("diagnostic library not built")
    L = capi.diag_lib()
    lds, grp, fits = C.c_uint64(0), C.c_int(0), C.c_int(0)
    assert L.uzl_debug_ml_admission(C.c_int(nb), C.c_int(n2), C.byref(lds), C.byref(grp), C.byref(fits)) == 0
    return lds.value, grp.value, fits.value


def test_lds_boundary_of_the_gather_level_vector():
    limit = 140 * 1024
    # 6 n2 doubles + 64 bytes: the largest n2 that fits is (limit - 64) / 48
    n2_max = (limit - 64) // 48
    assert 6 * n2_max == 17910                       # (round 5's cap of 18432 rows = 3072 aggregates lay ABOVE the LDS limit)
    lds, _, fits = _adm(32 * n2_max, n2_max)
    assert lds == 48 * n2_max + 64 <= limit and fits == 1
    lds, _, fits = _adm(32 * (n2_max + 1), n2_max + 1)
    assert lds > limit and fits == 0
    assert _adm(32 * 3072, 3072)[2] == 0             # 6 n2 = 18432


def test_partial_count_boundary_of_ml_spmv():
    # ml_spmv runs two half workgroups per 32-row level-2 aggregate: 8192 partials = 4096 aggregates = 131072 free vertices
    assert _adm(131072, 100)[1] == 8192 and _adm(131072, 100)[2] == 1
    assert _adm(131073, 100)[1] == 8194 and _adm(131073, 100)[2] == 0
    assert _adm(10000, 313) == (48 * 313 + 64, 626, 1)                  # BASELINE config 4
