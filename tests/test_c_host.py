"""The C11 host layer (include/deciphon_host.h): builds and exports its API on CPU; its database path
(tests/c/test_db_host.c: press -> .dcp -> read back, partition table, header checks, framing reader)
runs on the CPU, also under ASan/UBSan; its scan test (tests/c/test_scan_host.c: the reference's
protein_profile goldens, profile_reader + thread_run + scan_run_local over a pressed database) runs on
the GPU."""
import ctypes as C
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "deciphon-old_amd", "host")
HOST_SO = os.path.join(ROOT, "deciphon-old_amd", "libdeciphon_host.so")
LIBDIR = os.path.join(ROOT, "deciphon-old_amd")


def build_host():
    subprocess.check_call(["make", "-C", HOST_DIR, "-s"])


def build_c_test(tmp_path, name, extra=(), link_host=True):
    exe = str(tmp_path / name)
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", name + ".c"), "-o", exe, "-L", LIBDIR]
    cmd += list(extra)
    cmd += (["-ldeciphon_host"] if link_host else []) + ["-ldcp_hip", "-lm", "-fopenmp", "-Wl,-rpath," + LIBDIR]
    subprocess.check_call(cmd)
    return exe


def declared_functions():
    text = open(os.path.join(ROOT, "include", "deciphon_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#define[^\n]*(\\\n[^\n]*)*", "", text)
    text = re.sub(r'#ifdef __cplusplus\s*\n(extern "C" \{|\})\s*\n#endif', "", text)
    # drop every {...} block (struct bodies with their function-pointer members, inline function bodies):
    # what is left at file scope are the prototypes of exported functions
    depth, out = 0, []
    for ch in text:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        elif depth == 0:
            out.append(ch)
    text = "".join(out)
    text = re.sub(r"static inline[^;]*?\)\s*(?=\n)", "", text)  # the signature of a (now body-less) inline function
    protos = re.findall(r"\b([a-z][a-z0-9_]+)\s*\([^;()]*(?:\([^()]*\)[^;()]*)*\)\s*;", text)
    skip = {"imm_state_name", "sizeof", "assert"}  # a function TYPE (typedef) / operators
    return sorted(set(n for n in protos if n not in skip and not n.startswith("dcp_")))


def test_host_layer_builds_and_exports_its_api(dcp, tmp_path):
    build_host()
    names = declared_functions()
    lib = C.CDLL(HOST_SO)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the reference's entry points for this path, by their reference names (SURVEY.md 8b surface 1 + 2)
    assert {"protein_profile_init", "protein_profile_setup", "protein_profile_absorb", "protein_profile_sample",
            "protein_profile_decode", "protein_profile_pack", "protein_profile_unpack", "protein_profile_write_dot",
            "profile_init", "profile_del", "profile_unpack", "profile_typeid", "profile_null_dp", "profile_alt_dp",
            "protein_model_init", "protein_model_setup", "protein_model_add_node", "protein_model_add_trans",
            "protein_model_del", "standard_profile_init", "standard_profile_pack", "standard_profile_unpack",
            "profile_reader_setup", "profile_reader_npartitions", "profile_reader_partition_size",
            "profile_reader_nprofiles", "profile_reader_rewind_all", "profile_reader_rewind", "profile_reader_next",
            "profile_reader_end", "profile_reader_del", "protein_db_reader_open", "db_reader_close",
            "protein_db_writer_open", "protein_db_writer_pack_profile", "db_writer_close",
            "thread_init", "thread_setup_job", "thread_setup_seq", "thread_run", "prod_fwrite", "prod_fopen",
            "prod_fclose", "protein_match_write_func", "protein_codec_next", "protein_h3reader_init",
            "protein_h3reader_next", "protein_h3reader_del",
            "imm_task_new", "imm_task_reset", "imm_task_setup", "imm_dp_viterbi", "imm_prod", "imm_prod_reset",
            "imm_dp_trans_idx", "imm_dp_change_trans", "imm_dp_pack", "imm_dp_unpack", "imm_seq", "imm_subseq",
            "imm_abc_typeid_name", "imm_rnd", "imm_lprob_sample", "imm_lprob_normalize"} <= set(names)
    # the scan test compiles against the header (it needs a GPU to run)
    assert os.path.exists(build_c_test(tmp_path, "test_scan_host"))


def test_db_host_on_cpu(dcp, tmp_path):
    build_host()
    exe = build_c_test(tmp_path, "test_db_host")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert "all checks passed" in r.stdout


def test_db_host_under_sanitizers(dcp, tmp_path):
    """The host layer's own C files compiled with ASan + UBSan (+ leak check) into the test."""
    srcs = sorted(glob.glob(os.path.join(HOST_DIR, "*.c")))
    exe = build_c_test(tmp_path, "test_db_host",
                       extra=["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] + srcs,
                       link_host=False)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert "all checks passed" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


STUBS = os.path.join(ROOT, "tests", "c", "stubs")


def test_scan_run_adapter_compiles_against_the_scheduler_interfaces(dcp, tmp_path):
    """integration/scan_run_adapter.c (scan_run(job_id, num_threads), src/server/scan.h:6) builds with -Werror
    against stub declarations of deciphon/sched/api.h, sched/structs.h, job.h and file.h, and makes the
    scheduler calls of src/server/scan.c:215-269 and no others."""
    build_host()
    assert os.path.exists(build_c_test(tmp_path, "test_scan_run_adapter", extra=["-I", STUBS]))
    src = open(os.path.join(ROOT, "integration", "scan_run_adapter.c")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    called = set(re.findall(r"\b(api_[a-z_]+)\s*\(", src))
    assert called == {"api_get_scan_by_job_id", "api_get_db", "api_download_db", "api_scan_num_seqs",
                      "api_scan_next_seq", "api_increment_job_progress", "api_upload_prods_file", "api_set_job_state"}


@pytest.mark.gpu
def test_scan_run_adapter_on_gpu(tmp_path):
    build_host()
    exe = build_c_test(tmp_path, "test_scan_run_adapter", extra=["-I", STUBS])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert "all checks passed" in r.stdout


@pytest.mark.gpu
def test_c_scan_host_on_gpu(tmp_path):
    build_host()
    exe = build_c_test(tmp_path, "test_scan_host")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert "all checks passed" in r.stdout
