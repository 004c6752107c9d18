#!/bin/bash
# Row-sweep C3 step with each given build of libdcp_hip.so, twice, alternating: step rate and the ms of the multi-wavefront classes
#   gpurun -- "bash profiles/ab_rowsweep_mw.sh deciphon-old_amd/libdcp_hip.X.so deciphon-old_amd/libdcp_hip.Y.so"
for r in 1 2; do for v in "$@"; do cp $v deciphon-old_amd/libdcp_hip.so; python3 bench.py --kernel rowsweep --steps 2 --warmup 1 --no-cpu-baseline --e2e-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$v', d['value'], {k:r['per_class_ms_per_step'][k] for k in ('R3W4','R4W4','R3W8','R4W8') if k in r['per_class_ms_per_step']})"; done; done
