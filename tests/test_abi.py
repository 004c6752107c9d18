"""The C-ABI library loads and exports every symbol include/dcp_gpu.h declares.
No compute is launched here (CPU-only suite)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(hooks=False):
    text = open(os.path.join(ROOT, "include", "dcp_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    hook_blocks = re.findall(r"#ifdef DCP_TEST_HOOKS(.*?)#endif", text, flags=re.S)
    if hooks:
        text = "".join(hook_blocks)
    else:
        text = re.sub(r"#ifdef DCP_TEST_HOOKS.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dcp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(dcp):
    syms = declared_symbols()
    assert len(syms) >= 30
    lib = C.CDLL(dcp.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(dcp.ABI_SYMBOLS) == syms


def test_test_hooks_are_not_in_the_shipped_library(dcp):
    """Knobs on the result path (dcp_gpu_test_set_redo_cap) exist only in the tests' own
    -DDCP_TEST_HOOKS build, never in libdcp_hip.so or the host library."""
    hooks = declared_symbols(hooks=True)
    assert hooks == ["dcp_gpu_test_set_redo_cap", "dcp_gpu_test_set_ring_stall", "dcp_gpu_test_set_rowsweep_variant",
                     "dcp_gpu_test_set_seg_col_bytes", "dcp_gpu_test_set_trace_mode"]
    shipped = C.CDLL(dcp.LIB_PATH)
    assert not [h for h in hooks if hasattr(shipped, h)]
    host = os.path.join(os.path.dirname(dcp.LIB_PATH), "libdeciphon_host.so")
    if os.path.exists(host):
        assert not [h for h in hooks if hasattr(C.CDLL(host), h)]
    test_build = C.CDLL(dcp.TESTHOOKS_LIB_PATH)
    assert all(hasattr(test_build, h) for h in hooks)
    assert all(hasattr(test_build, s) for s in declared_symbols())  # otherwise the same library


def test_no_oracle_in_product():
    """The product tree must not reference the oracle (test infrastructure)."""
    bad = []
    prod = os.path.join(ROOT, "deciphon-old_amd")
    for d, _, files in os.walk(prod):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")) or f == "Makefile":
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"liboracle|oracle_py|oracle/|orc_[a-z]", txt):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_scanner_fails_loudly_without_gpu(dcp):
    if dcp.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(dcp.DcpError):
        dcp.Scanner(0)


def test_no_result_changing_environment_hooks():
    """Nothing in the shipped sources reads the environment: kernel choice goes
    through dcp_scan_params.kernel, the redo-list cap through a test-only setter, and the timing
    diagnostics of the query-lane kernel exist only as a compile-time -DDCP_QLANE_DIAG build."""
    bad = []
    for sub in ("csrc", "host"):
        d = os.path.join(ROOT, "deciphon-old_amd", sub)
        for f in os.listdir(d):
            if f.endswith((".cpp", ".hip", ".h", ".c")):
                if re.search(r"\bgetenv\b|\benviron\b", open(os.path.join(d, f), errors="ignore").read()):
                    bad.append(f)
    assert not bad, bad
