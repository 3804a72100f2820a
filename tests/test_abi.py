"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/kf.h declares, and rejects bad arguments without touching a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "kf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kf_[a-z_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from pymc_statespace_amd import _lib

    assert _header_functions() == sorted(_lib.EXPORTS)


def test_flag_bits_of_the_binding_match_the_header():
    from pymc_statespace_amd import _lib

    src = open(os.path.join(ROOT, "include", "kf.h")).read()
    bits = {name: int(sh) for name, sh in re.findall(r"\b(KF_FLAG_[A-Z_]+)\s*=\s*1u\s*<<\s*(\d+)", src)}
    assert bits["KF_FLAG_MVN_CONST_EXACT"] == 0 and _lib.KF_FLAG_MVN_CONST_EXACT == 1 << 0
    assert _lib.KF_FLAG_Y_PER_DRAW == 1 << bits["KF_FLAG_Y_PER_DRAW"]
    assert _lib.KF_FLAG_H_DIAGONAL == 1 << bits["KF_FLAG_H_DIAGONAL"]
    assert _lib.KF_FLAG_EXACT_RECURSION == 1 << bits["KF_FLAG_EXACT_RECURSION"]
    assert _lib.KF_FLAG_STEADY_STATE == 1 << bits["KF_FLAG_STEADY_STATE"]
    assert _lib.KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING == 1 << bits["KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING"]
    for name, bit in _lib.KF_FLAG_TV.items():
        assert bit == 1 << bits[f"KF_FLAG_TV_{name.upper()}"]
    assert len(set(bits.values())) == len(bits)  # no two flags share a bit


def test_library_exports_every_declared_symbol():
    from pymc_statespace_amd import _lib

    lib = _lib.load()
    for name in _header_functions():
        assert hasattr(lib, name), f"libkf_hip.so does not export {name}"
    assert lib.kf_abi_version() == _lib.KF_ABI_VERSION


def test_dims_struct_layout_matches_header():
    from pymc_statespace_amd._lib import KfDims

    # int64 B; int32 T,p,m,r,filter; uint32 flags  -> 8 + 6*4 = 32 bytes, no padding
    assert C.sizeof(KfDims) == 32
    assert KfDims.B.offset == 0 and KfDims.T.offset == 8 and KfDims.flags.offset == 28


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pymc_statespace_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libkf_hip.so"))
    with pytest.raises(_lib.KfLibraryMissing, match="no CPU fallback"):
        _lib.load()


def test_engine_refuses_to_run_without_gpu():
    import torch

    from pymc_statespace_amd import KalmanEngine, KalmanError

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(KalmanError, match="no CPU fallback"):
        KalmanEngine()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under pymc_statespace_amd/ may reference it."""
    pkg = os.path.join(ROOT, "pymc_statespace_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "kf_oracle" not in txt, f


def test_build_graph_needs_pytensor_and_says_so():
    """The symbolic entry point mirrors the reference signature; without PyTensor (absent from
    this image) it must fail with a clear ImportError, not fall back to anything."""
    import importlib.util

    from pymc_statespace_amd.filters import KalmanSmoother, StandardFilter

    if importlib.util.find_spec("pytensor") is not None:
        pytest.skip("PyTensor present")
    with pytest.raises(ImportError, match="needs PyTensor"):
        StandardFilter().build_graph(None, None, None, None, None, None, None, None)
    with pytest.raises(ImportError, match="needs PyTensor"):
        KalmanSmoother().build_graph(None, None, None, None, None)


def test_filter_factory_keys_match_reference():
    from pymc_statespace_amd.filters import FILTER_FACTORY

    # reference core/statespace.py:25-31
    assert set(FILTER_FACTORY) == {"standard", "univariate", "single", "cholesky", "steady_state"}
    for cls in FILTER_FACTORY.values():
        assert hasattr(cls, "build_graph") and callable(cls)


def test_op_layer_detects_time_varying_inputs_by_shape():
    """op._lead_system: a system matrix with one more axis than its static form is time-varying (reference
    filters/utilities.py:1-20); c / d are (m,1) columns when static and (n,m,1) when time-varying."""
    import numpy as np

    from pymc_statespace_amd import op

    n, m, p, r = 7, 3, 2, 2
    T, Z, R, H, Q = np.eye(m), np.ones((p, m)), np.ones((m, r)), np.eye(p), np.eye(r)
    c, d = np.zeros((m, 1)), np.zeros((p, 1))
    arrs, tv = op._lead_system(T, Z, R, H, Q, c, d)
    assert tv == () and [a.shape for a in arrs] == [(1, m, m), (1, p, m), (1, m, r), (1, p, p), (1, r, r), (1, m), (1, p)]
    Tt, Qt, ct = np.repeat(T[None], n, 0), np.repeat(Q[None], n, 0), np.zeros((n, m, 1))
    arrs, tv = op._lead_system(Tt, Z, R, H, Qt, ct, d)
    assert tv == ("T", "Q", "c")
    assert arrs[0].shape == (1, n, m, m) and arrs[4].shape == (1, n, r, r) and arrs[5].shape == (1, n, m) and arrs[6].shape == (1, p)


def test_steady_state_filter_has_a_symbolic_entry_and_a_gradient():
    """f4: SteadyStateFilter.build_graph exists again (PyTensor needed to call it) and the eager value-and-gradient is there."""
    from pymc_statespace_amd import filters

    assert callable(filters.SteadyStateFilter.build_graph) and callable(filters.SteadyStateFilter.logp_and_grad)
    with pytest.raises(ImportError):
        try:
            import pytensor  # noqa: F401
        except ImportError:
            filters.SteadyStateFilter(engine=object()).build_graph(None, None, None, None, None, None, None, None)
        else:
            raise ImportError("pytensor present: nothing to check here")


def test_the_dynamic_symbol_table_is_the_header():
    """-fvisibility=hidden (round 6): libkf_hip.so exports the entry points include/kf.h declares and nothing of its own making --
    what remains beside them are hipcc's per-translation-unit __hip_cuid_* markers and weak libstdc++ template instantiations."""
    import subprocess

    from pymc_statespace_amd import _lib

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    ours = {nm for nm in names if not nm.startswith("__hip_cuid_") and not nm.startswith("_ZNSt") and not nm.startswith("_ZSt")}
    assert ours == set(_lib.EXPORTS), sorted(ours ^ set(_lib.EXPORTS))


def test_option_table_is_the_set_of_switches_the_sources_read():
    """kf_set_option refuses names the library does not read; its table (csrc/kf_api.hip) must therefore be exactly the
    kf_env_switch("...") call sites of the sources -- and every one of them has a row in INTEGRATION.md's switch table."""
    import glob
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    src = ""
    for f in glob.glob(os.path.join(root, "pymc_statespace_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "pymc_statespace_amd", "csrc", "*.h")):
        src += open(f).read()
    read = set(re.findall(r'kf_env_switch\("([A-Z0-9_]+)"\)', src))
    api = open(os.path.join(root, "pymc_statespace_amd", "csrc", "kf_api.hip")).read()
    tbl = re.search(r"kf_option_names\[\] = \{(.*?)\};", api, re.S).group(1)
    table = set(re.findall(r'"([A-Z0-9_]+)"', tbl))
    assert table == read, (sorted(table - read), sorted(read - table))
    assert all(len(n) < 32 for n in table)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in table if n not in doc)
    assert not missing, f"switches without a row in INTEGRATION.md: {missing}"
