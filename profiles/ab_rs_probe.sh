#!/bin/bash
# Row-sweep latency points of several builds on one box:
#   gpurun -- 'bash profiles/ab_rs_probe.sh deciphon-old_amd/libdcp_hip.A.so deciphon-old_amd/libdcp_hip.B.so ...'
set -e
for v in "$@"; do
  cp "$v" deciphon-old_amd/libdcp_hip.so
  echo -n "$v  "; timeout -k 10 300 python3 profiles/rs_probe.py $RS_NQ 2>/dev/null | tail -1
done
