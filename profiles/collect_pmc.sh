#!/bin/bash
# Collect the rocprofv3 evidence for one bench.py command on the GPU box:
#   gpurun --timeout 1100 -- 'bash profiles/collect_pmc.sh TAG [bench.py args...]'
# pass 0  --kernel-trace --stats            (per-kernel durations)
# pass 1+ --pmc ... one counter group each  (separate runs: FETCH_SIZE and WRITE_SIZE do not
#         fit one pass; SQ has 8 slots -- MI355X_MICROARCH.md, rocprofv3 PMC slots)
# Output under gpurun_out/TAG/; fold with profiles/pmc_summary.py, copy what is judged into
# profiles/rNN/.  The program after `--` is python3 itself (no env/bash hop under the profiler).
set -e -o pipefail
TAG=${1:?tag}; shift
ARGS=${*:---steps 1 --warmup 0 --no-cpu-baseline}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() { # name, rocprof options...
    local name=$1; shift
    echo "== pass $name: $*"
    timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
    tail -c 300 "$OUT/$name.json"; echo
}
# PASSES selects a subset (default: all five), so that a long command fits gpurun's time limit in two calls:
#   PASSES="stats fetch write" ... ; PASSES="sq lds" ...   (same TAG: the summary folds what is there)
PASSES=${PASSES:-stats fetch write sq lds}
for p in $PASSES; do
  case $p in
    stats) run stats --kernel-trace --stats ;;
    fetch) run pmc_fetch --pmc FETCH_SIZE ;;
    write) run pmc_write --pmc WRITE_SIZE ;;
    sq)    run pmc_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE ;;
    lds)   run pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM ;;
    l2)    run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum ;;   # not in the default set: PASSES="... l2"
  esac
done
cd "$ROOT"
DIRS=""
for d in pmc_fetch pmc_write pmc_sq pmc_lds pmc_l2; do [ -d "$OUT/$d" ] && DIRS="$DIRS $OUT/$d"; done
[ -n "$DIRS" ] && python3 profiles/pmc_summary.py "$OUT/pmc.json" $DIRS > "$OUT/pmc_derived.txt" && cat "$OUT/pmc_derived.txt"
[ -d "$OUT/stats" ] && cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
true
