#!/bin/bash
# Row-sweep C3 step with each given build of libdcp_hip.so, twice, alternating: step rate and every class's ms per step
#   gpurun -- "bash profiles/ab_rowsweep_all.sh deciphon-old_amd/libdcp_hip.X.so deciphon-old_amd/libdcp_hip.Y.so"
for r in 1 2; do for v in "$@"; do cp $v deciphon-old_amd/libdcp_hip.so; python3 bench.py --kernel rowsweep --steps 2 --warmup 1 --no-cpu-baseline --e2e-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$v', d['value'], r['per_class_ms_per_step'])"; done; done
for v in "$@"; do cp $v deciphon-old_amd/libdcp_hip.so; python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --e2e-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v small_batches', {k:v['ms'] for k,v in d['small_batches'].items() if k!='what'})"; done
