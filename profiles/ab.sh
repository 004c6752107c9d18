#!/bin/bash
# A/B two builds of libdcp_hip.so on the same GPU box (alternating, so clock / box drift cancels):
#   gpurun -- 'bash profiles/ab.sh deciphon-old_amd/libdcp_hip.A.so deciphon-old_amd/libdcp_hip.B.so [rounds] [bench args]'
set -e
A=$1; B=$2; R=${3:-2}; shift 3 || true
ARGS=${*:---steps 2 --warmup 1 --no-cpu-baseline}
for r in $(seq 1 $R); do
  for v in "$A" "$B"; do
    cp "$v" deciphon-old_amd/libdcp_hip.so
    echo -n "$v: "
    timeout -k 10 300 python bench.py $ARGS | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['per_class_ms_per_step'].get('qlane_KT8'))"
  done
done
