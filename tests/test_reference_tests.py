"""The reference's own tests, compiled UNCHANGED against this library's headers.

oracle/compat_tests.mk compiles /root/reference/test/protein_model.c and test/protein_profile.c where
they lie -- no source is copied, only the `-I include/compat` forwarding headers stand between them
and include/deciphon_host.h -- into oracle/_ref/compat_tests/ (git-ignored; the binaries travel to
the GPU box, the reference does not).  protein_model runs on the CPU; protein_profile's
imm_dp_viterbi calls need the MI355X and check the goldens G1-G3 of test/protein_profile.c:41,65,157
with the reference's own CLOSE tolerance (5e-5 relative for float32, test/hope_support.h:26).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "oracle", "_ref", "compat_tests")


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "deciphon-old_amd", "host"), "-s"])
    subprocess.check_call(["make", "-s", "-f", os.path.join(ROOT, "oracle", "compat_tests.mk")])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_reference_tests_compile_unchanged_and_model_test_passes(dcp):
    build()
    for name in ("protein_model", "protein_profile"):
        assert os.access(os.path.join(OUT, name), os.X_OK)
    r = subprocess.run([os.path.join(OUT, "protein_model")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr


def test_compat_headers_only_forward():
    """include/compat/ holds no declarations of its own: every file is a comment + one #include."""
    top = os.path.join(ROOT, "include", "compat")
    n = 0
    for d, _, files in os.walk(top):
        for f in files:
            code = [l.strip() for l in open(os.path.join(d, f)) if l.strip() and not l.strip().startswith(("/*", "*"))]
            assert code == ['#include "deciphon_host.h"'], (f, code)
            n += 1
    assert n >= 30


@pytest.mark.gpu
def test_reference_protein_profile_test_passes_on_gpu():
    exe = os.path.join(OUT, "protein_profile")
    if not os.path.exists(exe):
        if not os.path.isdir(REF):
            pytest.skip("oracle/_ref/compat_tests/protein_profile was not built (no reference tree at build time)")
        build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Assertion error" not in r.stderr
