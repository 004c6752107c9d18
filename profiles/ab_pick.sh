#!/bin/bash
# A/B two builds on the row-sweep C3 step (profiles/ab_rowsweep_mw.sh), then install the one with the lower R3W4 time
# as libdcp_hip.so and say which:   gpurun -- 'bash profiles/ab_pick.sh deciphon-old_amd/libdcp_hip.X.so deciphon-old_amd/libdcp_hip.Y.so'
A=$1; B=$2
mkdir -p gpurun_out
bash profiles/ab_rowsweep_mw.sh $A $B | tee gpurun_out/ab_pick.txt
python3 - "$A" "$B" <<'PY'
import re,sys,shutil
a,b=sys.argv[1:3]
t={a:[],b:[]}
for l in open('gpurun_out/ab_pick.txt'):
    m=re.match(r"(\S+) \S+ \{'R3W4': ([0-9.]+)",l)
    if m: t[m.group(1)].append(float(m.group(2)))
ma,mb=sum(t[a])/len(t[a]),sum(t[b])/len(t[b])
w=a if ma<=mb else b
print("R3W4 ms:",a,ma,b,mb,"-> installing",w)
shutil.copy(w,'deciphon-old_amd/libdcp_hip.so')
open('gpurun_out/ab_pick_winner.txt','w').write(w+"\n")
PY
