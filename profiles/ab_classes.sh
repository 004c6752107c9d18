#!/bin/bash
# Like ab.sh, but prints the per-size-class launch times of each build (row-sweep tuning):
#   gpurun -- 'bash profiles/ab_classes.sh ROUNDS deciphon-old_amd/libdcp_hip.A.so deciphon-old_amd/libdcp_hip.B.so ...'
set -e
R=$1; shift
ARGS=${BENCH_ARGS:---kernel rowsweep --steps 2 --warmup 1 --no-cpu-baseline}
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp "$v" deciphon-old_amd/libdcp_hip.so
    timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['value'], {k: round(x, 1) for k, x in d['roofline']['per_class_ms_per_step'].items()})"
  done
done
