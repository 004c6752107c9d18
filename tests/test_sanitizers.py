"""ASan + UBSan on the CPU builds (upstream's CI runs its tests under both:
.github/workflows/test.yml:27).  GPU AddressSanitizer is not available on this pool, so the device
code is covered by bit-exact parity instead."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "check-asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("oracle selftest ok") == 2


def test_host_model_under_asan_ubsan(tmp_path):
    from test_h3reader import random_model, write_hmm

    rng = np.random.default_rng(4)
    good, bad = tmp_path / "two.hmm", tmp_path / "bad.hmm"
    write_hmm(good, [("a", "PF1.1", *random_model(rng, 3)), ("b", "", *random_model(rng, 40))])
    bad.write_text(good.read_text().replace("ALPH  amino", "ALPH  RNA"))
    exe = str(tmp_path / "asan_host_model")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "deciphon-old_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "asan_host_model.cpp"),
                           os.path.join(ROOT, "deciphon-old_amd", "csrc", "dcp_model.cpp"), "-o", exe])
    r = subprocess.run([exe, str(good), str(bad)], capture_output=True, text=True,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "asan_host_model ok" in r.stdout


def test_file_parsers_survive_mutation_fuzzing_under_asan_ubsan(dcp, tmp_path):
    """tests/c/fuzz_parsers.cpp: thousands of mutated HMMER3 and dcpx files (byte edits, bit flips,
    truncations, duplicated and deleted runs) through dcp_h3reader_* / dcp_db_* under ASan + UBSan +
    LeakSanitizer: clean error codes only."""
    from test_h3reader import random_model, write_hmm

    rng = np.random.default_rng(5)
    hmm, dbx = tmp_path / "good.hmm", tmp_path / "good.dcpx"
    write_hmm(hmm, [("a", "PF1.1", *random_model(rng, 3)), ("b", "", *random_model(rng, 17))])
    dcp.write_db(str(dbx), dcp.read_hmmer3(str(hmm)))
    exe = str(tmp_path / "fuzz_parsers")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "deciphon-old_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "fuzz_parsers.cpp"),
                           os.path.join(ROOT, "deciphon-old_amd", "csrc", "dcp_model.cpp"), "-o", exe])
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    r = subprocess.run([exe, str(hmm), str(dbx), str(scratch), "4000", "20261004"], capture_output=True, text=True,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fuzz_parsers ok" in r.stdout


def test_dcp_database_path_survives_mutation_fuzzing_under_asan_ubsan(dcp, tmp_path):
    """tests/c/fuzz_dcp.c: thousands of mutated .dcp files (byte edits, truncations, duplicated and deleted
    runs, blown-up length fields, bit flips) through protein_db_reader_open + profile_reader_setup(_balanced)
    + profile_reader_next, the host layer's own C files built with ASan + UBSan + leak check: every file ends
    in RC_END or a clean error code."""
    import glob
    host = sorted(glob.glob(os.path.join(ROOT, "deciphon-old_amd", "host", "*.c")))
    exe = str(tmp_path / "fuzz_dcp")
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "fuzz_dcp.c")] + host +
                          ["-o", exe, "-L", os.path.join(ROOT, "deciphon-old_amd"), "-ldcp_hip", "-lm", "-fopenmp",
                           "-Wl,-rpath," + os.path.join(ROOT, "deciphon-old_amd")])
    log = str(tmp_path / "asan")
    env = dict(os.environ, ASAN_OPTIONS=f"detect_leaks=1:log_path={log}")
    r = subprocess.run([exe, str(tmp_path / "fuzz.dcp"), "6000", "20261004"], capture_output=True, text=True,
                       timeout=600, env=env)
    reports = [f for f in os.listdir(tmp_path) if f.startswith("asan")]
    assert r.returncode == 0 and not reports, r.stdout + "".join(open(os.path.join(tmp_path, f)).read()[-2000:] for f in reports)
    assert "fuzz_dcp ok" in r.stdout
