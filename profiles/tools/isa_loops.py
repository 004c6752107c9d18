#!/usr/bin/env python3
"""Inner loops of a kernel in hipcc's -S output: instructions, VALU count, static s_waitcnt vmcnt(0), spill traffic.

    hipcc --offload-arch=gfx950 <Makefile's HIPFLAGS> -S --cuda-device-only dcp_qlane.hip -o q.s
    python3 profiles/tools/isa_loops.py q.s viterbi_qlane2_kernel [min_instructions]

The workflow check of DESIGN.md 4.3a: the five-row loops of the query-lane sweeps must hold no `s_waitcnt vmcnt(0)`
(it drains the three-row boundary prefetch), no scratch_load / scratch_store and no v_readlane / v_writelane
(SGPR spills) -- a loop = the lines between a label and the LAST backward branch to it.
"""
import re
import sys


def kernel_lines(path, name):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"_Z\d+" + name, l) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def loops(lines):
    label_at = {}
    for i, l in enumerate(lines):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    out = {}
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in label_at and label_at[tgt] < i:
                out[tgt] = max(out.get(tgt, 0), i)
    return sorted((label_at[t], e, t) for t, e in out.items())


def vmcnt0(l):
    m = re.search(r"s_waitcnt\s+(.*)", l)
    return bool(m and re.search(r"vmcnt\(0\)", m.group(1)))


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    lines = kernel_lines(path, name)
    ls = loops(lines)
    # the row loops: loops that contain no other loop but small ones (the ring hand-shake's polling loops)
    for b, e, t in ls:
        if any(b < b2 and e2 < e and e2 - b2 > 100 for b2, e2, _ in ls):
            continue
        body = [l.strip() for l in lines[b:e + 1] if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
        if len(body) < min_ins:
            continue
        valu = sum(1 for l in body if l.startswith("v_") and not l.startswith(("v_readlane", "v_writelane", "v_readfirstlane")))
        print("%s: %d instructions, %d VALU, vmcnt(0) waits %d, scratch ops %d, readlane/writelane %d, ds ops %d, global ld/st %d/%d"
              % (t, len(body), valu, sum(vmcnt0(l) for l in body), sum(l.startswith("scratch_") for l in body),
                 sum(l.startswith(("v_readlane", "v_writelane")) for l in body), sum(l.startswith("ds_") for l in body),
                 sum(l.startswith("global_load") for l in body), sum(l.startswith("global_store") for l in body)))


if __name__ == "__main__":
    main()
