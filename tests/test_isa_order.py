"""The two-stage query-lane kernel hands boundary rows from one wavefront to its partner through an LDS ring
whose flags are RELAXED workgroup atomics: the order data -> flag (producer) and flag -> data -> "taken"
(consumer) is not an edge of the HIP memory model, it relies on (a) the compiler emitting the DS operations
in program order across the fences and (b) the LDS executing one wavefront's DS operations in issue order
(DESIGN.md 4.3).  (a) is checked here on every build: the kernel source leaves comment marks in
the ISA around each hand-off and this test assembles the file and inspects what lies between them."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deciphon-old_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def qlane2_isa(tmp_path_factory):
    if not shutil.which(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "qlane.s"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
             "-ffp-contract=off", "-fno-honor-nans", "-S", "--cuda-device-only"]  # the Makefile's HIPFLAGS
    subprocess.run([HIPCC] + flags + [os.path.join(CSRC, "dcp_qlane.hip"), "-o", str(out)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = out.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z21viterbi_qlane2_kernel"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def spans(lines, begin, end):
    """Instruction mnemonics between every `; begin` mark and the next `; end` mark."""
    out, cur = [], None
    for l in lines:
        t = l.strip()
        if t == "; " + begin:
            cur = []
        elif t == "; " + end and cur is not None:
            out.append(cur)
            cur = None
        elif cur is not None and t and not t.startswith((";", ".")) and not t.endswith(":"):
            assert not t.startswith(("s_cbranch", "s_branch")), f"control flow between {begin} and {end}"
            cur.append(t.split()[0])
    return out


def ds(ops):
    return [o for o in ops if o.startswith("ds_")]


@pytest.mark.timeout(600)
def test_ring_hand_off_order_in_the_emitted_isa(qlane2_isa):
    data = spans(qlane2_isa, "DCP_RING_DATA", "DCP_RING_FLAG")
    flag = spans(qlane2_isa, "DCP_RING_FLAG", "DCP_RING_DATA_END")
    assert len(data) == len(flag) and len(data) >= 10  # 2 producer sweep variants x (5 unrolled rows + 4 tail rows)
    for d, f in zip(data, flag):
        assert ds(d) == ["ds_write_b32"] * 3, d      # Xm, Xd, E of the row ...
        assert ds(f) == ["ds_write_b32"], f          # ... and only then the row number
    take = spans(qlane2_isa, "DCP_RING_TAKE", "DCP_RING_TAKEN")
    taken = spans(qlane2_isa, "DCP_RING_TAKEN", "DCP_RING_TAKE_END")
    assert len(take) == len(taken) and len(take) >= 10
    for t, k in zip(take, taken):
        # the row's three values (fewer where the last rows of a sweep no longer use them: a dead load is dropped) ...
        assert len(ds(t)) <= 3 and all(o == "ds_read_b32" for o in ds(t)), t
        assert ds(k) == ["ds_write_b32"], k          # ... before the slot is given back
    assert sum(1 for t in take if ds(t) == ["ds_read_b32"] * 3) >= 10
    # a wait for the partner ends with the flag load; no ring data is read before that mark: between an
    # ACQUIRED mark and the TAKE mark that follows it there is no DS read at all
    text = "\n".join(l.strip() for l in qlane2_isa)
    for m in re.finditer(r"; DCP_RING_ACQUIRED\n(.*?); DCP_RING_TAKE\n", text, flags=re.S):
        between = [l.split()[0] for l in m.group(1).splitlines() if l and not l.startswith((";", ".")) and not l.endswith(":")]
        if any(o.startswith(("s_cbranch", "s_branch")) for o in between):
            continue  # the next TAKE belongs to another path
        assert not [o for o in between if o.startswith("ds_read_b32")], between
    assert text.count("; DCP_RING_ACQUIRED") >= 4
