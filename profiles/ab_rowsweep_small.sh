#!/bin/bash
# Row-sweep C3 step with each given build of libdcp_hip.so, twice, alternating: step rate, ms of the two smallest classes,
# and the small_batches leg of a short automatic run (1 / 8 / 64 queries)
#   gpurun -- "bash profiles/ab_rowsweep_small.sh deciphon-old_amd/libdcp_hip.X.so deciphon-old_amd/libdcp_hip.Y.so"
for r in 1 2; do for v in "$@"; do cp $v deciphon-old_amd/libdcp_hip.so; python3 bench.py --kernel rowsweep --steps 2 --warmup 1 --no-cpu-baseline --e2e-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$v', d['value'], {k:(r['per_class_ms_per_step'][k], r['per_class_gcells_per_s'][k]) for k in ('R1W1','R2W1','R3W1','R4W1')})"; done; done
for v in "$@"; do cp $v deciphon-old_amd/libdcp_hip.so; python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --e2e-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v small_batches', {k:v['ms'] for k,v in d['small_batches'].items() if k!='what'})"; done
