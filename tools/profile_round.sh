#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 evidence of one round, run ON THE GPU BOX:
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01_v4'
# Kernel-trace statistics of the bench command (c2 and c3) and, in SEPARATE passes, the HBM counters
# (FETCH_SIZE / WRITE_SIZE) and the SQ activity counters of the c2 command.  Raw output lands in
# gpurun_out/<TAG>/; tools/summarize_profiles.py condenses it into the files kept under profiles/.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 10 --warmup 2"
$BENCH > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
$BENCH --workload c3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
$BENCH --workload c2g --no-cpu-baseline > "$OUT/bench_c2g.json" 2> "$OUT/bench_c2g.err"
$BENCH --workload c3g --no-cpu-baseline > "$OUT/bench_c3g.json" 2> "$OUT/bench_c3g.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c2" -o c2 -- $BENCH --no-cpu-baseline > "$OUT/trace_c2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -o c3 -- $BENCH --no-cpu-baseline --workload c3 > "$OUT/trace_c3.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- $BENCH --no-cpu-baseline --steps 4 --warmup 1 > "$OUT/pmc_$C.log" 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_SQ" -o pmc -- $BENCH --no-cpu-baseline --steps 4 --warmup 1 > "$OUT/pmc_SQ.log" 2>&1
python "$OLDPWD/tools/summarize_profiles.py" "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
