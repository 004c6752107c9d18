#!/bin/bash
# Next step for the one-layout DB: price the gather with four loads in flight per pass
# (profiles/r04/one_layout/unrolled_gather_not_measured.patch) against the shipped one, and against the two-layout DB.
# Run HERE (CPU container) to build the two libraries, then hand the printed command to gpurun:
#   bash profiles/ab_one_layout.sh build
# A = the tree as it is, B = the tree + the patch (the patch is reverted again; nothing stays applied).
# PATCH=profiles/r04/one_layout/unrolled_tile_loads_not_measured.patch builds B with four loads in flight in the
# DEFAULT kernels' copy of a stored tile image as well (the shipped copy has two: one load per iteration behind a
# vmcnt(1)) -- that one touches the headline kernel: price it on the plain driver workload too
# (BENCH_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --e2e-steps 0" bash profiles/ab.sh 3 A B) and run the whole
# `pytest -m gpu` on B before keeping it.
set -e
cd "$(dirname "$0")/.."
case ${1:-build} in
build)
  make -C deciphon-old_amd/csrc -j6 >/dev/null
  cp deciphon-old_amd/libdcp_hip.so deciphon-old_amd/libdcp_hip.A.so
  PATCH=${PATCH:-profiles/r04/one_layout/unrolled_gather_not_measured.patch}
  git apply "$PATCH"
  trap 'git apply -R "$PATCH"; make -C deciphon-old_amd/csrc -j6 >/dev/null' EXIT
  make -C deciphon-old_amd/csrc -j6 >/dev/null
  cp deciphon-old_amd/libdcp_hip.so deciphon-old_amd/libdcp_hip.B.so
  cp deciphon-old_amd/libdcp_hip_testhooks.so deciphon-old_amd/libdcp_hip_testhooks.B.so
  cat <<'EOF'
built deciphon-old_amd/libdcp_hip.{A,B}.so.  On the GPU (B must pass the parity test before its time means anything):
  gpurun --timeout 900 -- 'cp deciphon-old_amd/libdcp_hip.B.so deciphon-old_amd/libdcp_hip.so && cp deciphon-old_amd/libdcp_hip_testhooks.B.so deciphon-old_amd/libdcp_hip_testhooks.so && python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_table_layout or packed_slots or kernels_agree" && BENCH_ARGS="--one-layout --qlen 100 --qstep 10000 --steps 2 --warmup 1 --no-cpu-baseline --e2e-steps 0" bash profiles/ab.sh 3 deciphon-old_amd/libdcp_hip.A.so deciphon-old_amd/libdcp_hip.B.so && BENCH_ARGS="--one-layout --steps 2 --warmup 1 --no-cpu-baseline --e2e-steps 0" bash profiles/ab.sh 2 deciphon-old_amd/libdcp_hip.A.so deciphon-old_amd/libdcp_hip.B.so'
(two layouts on the same workloads: the same ab.sh lines without --one-layout.)
EOF
  ;;
*) echo "usage: $0 build"; exit 2;;
esac
