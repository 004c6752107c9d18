#!/bin/bash
# Regenerates the round's bench-line evidence on one GPU box (outputs under gpurun_out/$TAG/; copy what is
# judged into profiles/rNN/).  Two parts so that each fits one gpurun call:
#   gpurun --timeout 1100 -- 'bash profiles/measure_round.sh r02m driver'
#   gpurun --timeout 1100 -- 'bash profiles/measure_round.sh r02m lines'
#   gpurun --timeout 1100 -- 'bash profiles/measure_round.sh r04m pmc'
#   gpurun --timeout 1190 -- 'bash profiles/measure_round.sh r03m c5full'
set -o pipefail
TAG=${1:?tag}; PART=${2:?driver|lines|pmc|c5full}
OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { # name, bench args...
  local name=$1; shift
  timeout -k 10 ${LINE_TIMEOUT:-300} python3 bench.py "$@" 2> $OUT/$name.err | tail -1 > $OUT/$name.json
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', d['value'], d['unit'], 'ms/step', d['ms_per_step'])" || { echo "$name FAILED"; tail -5 $OUT/$name.err; }
}
case $PART in
driver)
  line c3_driver_command_bench --steps 20 --warmup 5
  PASSES=stats PASS_TIMEOUT=500 bash profiles/collect_pmc.sh $TAG/driver_rocprof --steps 20 --warmup 5
  ;;
lines)
  line c3_rowsweep_bench --kernel rowsweep --steps 3 --warmup 1 --no-cpu-baseline
  line c3_qlane_single_stage_bench --kernel qlane --steps 3 --warmup 1 --no-cpu-baseline
  line c2_bench --workload c2 --steps 5 --warmup 2 --no-cpu-baseline
  line c5_bench_1000q --workload c5 --qstep 1000 --steps 3 --warmup 1 --no-cpu-baseline --e2e-steps 0
  line c5_bench_8192q_4000prof --workload c5 --nprof 4000 --qstep 8192 --steps 2 --warmup 1 --no-cpu-baseline --e2e-steps 0
  line c3sizes_dense10_consensus_queries_bench --dense 10 --steps 3 --warmup 1 --no-cpu-baseline
  line c3sizes_dense10_random_queries_bench --dense 10 --dense-random --steps 3 --warmup 1 --no-cpu-baseline
  line c3_planted_bench --planted --steps 3 --warmup 1 --no-cpu-baseline
  DCP_BENCH_FORCE_DIST=1 line c3_bench_c_rccl_gather_1rank --steps 3 --warmup 1 --no-cpu-baseline
  timeout -k 10 400 python3 profiles/latency_probe.py > $OUT/latency_probe.txt 2>&1; grep "^auto" $OUT/latency_probe.txt
  timeout -k 10 300 python3 profiles/smalldb_probe.py > $OUT/smalldb_probe.txt 2>&1; tail -3 $OUT/smalldb_probe.txt
  ;;
pmc)
  # the headline's literal 10 000-query batch, the HBM counter passes of the driver's workload
  # (FETCH_SIZE / WRITE_SIZE in separate runs; folded by pmc_summary.py), and the row sweep's per-class counters
  line c3_bench_10000q --qstep 10000 --steps 2 --warmup 1 --no-cpu-baseline --e2e-steps 0
  PASSES="fetch write" PASS_TIMEOUT=300 bash profiles/collect_pmc.sh $TAG/pmc_qlane2 --steps 5 --warmup 1 --no-cpu-baseline --e2e-steps 0
  timeout -k 10 400 python3 profiles/latency_probe.py c5 > $OUT/latency_probe_c5_db.txt 2>&1; grep "^auto" $OUT/latency_probe_c5_db.txt
  ;;
c5full)
  # BASELINE configs[4] at full size on one GPU: all 20 000 profiles (M 50-2000) x one step of 8 192 mixed-length
  # queries (1.9e14 cells: ~2 min per step)
  LINE_TIMEOUT=1000 line c5_bench_8192q_full_db --workload c5 --qstep 8192 --steps 1 --warmup 1 --no-cpu-baseline --e2e-steps 0
  ;;
esac
true
