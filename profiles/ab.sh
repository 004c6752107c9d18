#!/bin/bash
# Compare builds of libdcp_hip.so on the same GPU box (alternating, so clock / box drift cancels):
#   gpurun -- 'bash profiles/ab.sh ROUNDS deciphon-old_amd/libdcp_hip.A.so deciphon-old_amd/libdcp_hip.B.so ...'
# The last variant stays installed as libdcp_hip.so.  BENCH_ARGS overrides the bench command line.
set -e
R=$1; shift
ARGS=${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline}
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp "$v" deciphon-old_amd/libdcp_hip.so
    timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['value'], d['roofline']['per_class_ms_per_step'].get('qlane_KT8'))"
  done
done
