"""The C-ABI library loads without a GPU and exports every entry point include/cone_engine.h declares; argument
validation that needs no device works.  (No compute calls here: those are the -m gpu tests.)"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cvxpylayers_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cone_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ce_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    for sym in declared_symbols():
        assert hasattr(L, sym), sym


def test_struct_layouts_agree_between_library_binding_and_the_documented_stub():
    """ce_template / ce_settings carry no size field: the library reports sizeof() of its own structs (ce_struct_size) and both the
    binding (_lib.py) and the stub a maintainer would copy out of INTEGRATION.md section 2 must match it field by field."""
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "cone_engine.h")).read()
    assert int(re.search(r"#define CE_ABI_VERSION (\d+)", hdr).group(1)) == L.ce_abi_version() == _lib.ABI_VERSION
    assert L.ce_struct_size(0) == C.sizeof(_lib.CeTemplate) and L.ce_struct_size(1) == C.sizeof(_lib.CeSettings)
    assert L.ce_struct_size(2) == -1
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    classes = re.findall(r"^class (\w+)\(C\.Structure\):.*\n((?:    .*\n)+)", doc, flags=re.M)
    ns = {"C": C}
    for name, body in classes:
        exec(f"class {name}(C.Structure):\n{body}", ns)
    for stub, mine in (("ce_template", _lib.CeTemplate), ("ce_settings", _lib.CeSettings)):
        assert stub in ns, f"INTEGRATION.md no longer documents {stub}"
        assert [(f, t) for f, t in ns[stub]._fields_] == [(f, t) for f, t in mine._fields_], stub
        assert C.sizeof(ns[stub]) == C.sizeof(mine)
    # every field of the header's structs, in order (catches a field added to the header but to neither python struct)
    for cname, mine in (("ce_template", _lib.CeTemplate), ("ce_settings", _lib.CeSettings)):
        body = re.search(r"typedef struct \{([^{}]*)\} " + cname + ";", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S), flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                names += [re.sub(r"[\s\*]", "", part).split(" ")[-1] for part in re.sub(r"^(const\s+)?(int|double)\s+", "", decl).split(",")]
        assert names == [f for f, _ in mine._fields_], (cname, names)
    assert f"ce_abi_version() == {_lib.ABI_VERSION}" in doc


def test_default_settings_are_scs_defaults():
    s = _lib.CeSettings()
    _lib.lib().ce_default_settings(C.byref(s))
    assert (s.eps_abs, s.eps_rel, s.eps_infeas, s.alpha, s.rho_x, s.scale) == (1e-4, 1e-4, 1e-7, 1.5, 1e-6, 0.1)
    assert (s.max_iters, s.normalize, s.adaptive_scale) == (100000, 1, 1)


def test_bad_template_is_rejected_before_touching_a_device():
    L = _lib.lib()
    t = _lib.CeTemplate()
    idx = np.zeros(1, dtype=np.int32)
    t.n, t.m, t.nnz_aug = 0, 0, 0
    t.indices = idx.ctypes.data_as(C.POINTER(C.c_int)); t.indptr = idx.ctypes.data_as(C.POINTER(C.c_int))
    h = C.c_void_p()
    assert L.ce_create(C.byref(t), 0, C.byref(h)) == -1      # CE_E_BADARG
    assert b"template" in L.ce_last_error()
    assert L.ce_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libcone_engine.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_plugin_refuses_cpu_only_host():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cvxpylayers_amd import problems as P
    from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer
    tpl = P.dense_template(3, {"z": 0, "l": 4, "q": []})
    ctx = MI355_ctx(None, tpl.problem_data_index, tpl.cones)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _CvxpyLayer.apply(None, torch.zeros(4, 2, dtype=torch.double), torch.zeros(tpl.nnz_aug, 2, dtype=torch.double), ctx, {}, False, None)


def test_solver_args_map_to_ce_settings_like_diffcp_maps_them_to_scs():
    """diffcp maps `eps` to eps_abs and eps_rel; SCS option names pass through; options that only steer diffcp's CPU execution are
    accepted and ignored; anything unknown is an error (a typo must not silently run with defaults)."""
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    s = make_settings({"eps": 1e-7, "max_iters": 123, "alpha": 1.2, "scale": 0.5, "normalize": False, "adaptive_scale": 0,
                       "acceleration_lookback": 10, "n_jobs_forward": 4, "mode": "lsqr", "verbose": True})
    assert s.eps_abs == 1e-7 and s.eps_rel == 1e-7 and s.max_iters == 123 and s.alpha == 1.2 and s.scale == 0.5
    assert s.normalize == 0 and s.adaptive_scale == 0 and s.warm_start == 0
    assert s.acceleration_lookback == 10 and s.acceleration_interval == 10
    s2 = make_settings({"eps_abs": 1e-5, "eps_rel": 1e-3})
    assert s2.eps_abs == 1e-5 and s2.eps_rel == 1e-3
    # one contract at every layer: ce_default_settings = SCS 3 defaults (which diffcp forwards, diffcp_if.py:356-367), acceleration included
    assert s2.acceleration_lookback == 10 and s2.acceleration_interval == 10 and s2.alpha == 1.5 and s2.rho_x == 1e-6 and s2.scale == 0.1 and s2.eps_infeas == 1e-7
    assert make_settings({"acceleration_lookback": 0}).acceleration_lookback == 0
    with pytest.raises(ValueError, match="unknown solver_args"):
        make_settings({"epsilon": 1e-3})


def test_cone_dims_objects_and_dicts_give_the_same_solver_dict():
    """cone_dims arrives as CVXPY's ConeDims (attrs zero / nonneg / soc / exp / psd / p3d) or as an SCS-style dict (keys z or f, l, q, ep, s, p)"""
    from types import SimpleNamespace
    from cvxpylayers_amd.interfaces.mi355_if import dims_to_solver_dict
    obj = SimpleNamespace(zero=2, nonneg=3, soc=[4, 5], exp=1, psd=[3], p3d=[0.5])
    want = {"z": 2, "l": 3, "q": [4, 5], "ep": 1, "s": [3], "p": [0.5]}
    assert dims_to_solver_dict(obj) == want
    assert dims_to_solver_dict({"f": 2, "l": 3, "q": [4, 5], "ep": 1, "s": [3], "p": [0.5]}) == want
    assert dims_to_solver_dict({"z": 1}) == {"z": 1, "l": 0, "q": [], "ep": 0, "s": [], "p": []}
