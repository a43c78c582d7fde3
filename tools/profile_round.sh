#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 evidence of one round, run ON THE GPU BOX:
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r02'
# Bench lines of every workload; kernel-trace statistics of c2 / c3 / the B = 65536 throughput regime / the
# receiver function alone / the chain workloads; and, in SEPARATE passes (never together with a trace domain),
# the HBM counters (FETCH_SIZE / WRITE_SIZE) and the SQ activity counters of the c2 and c3 commands.
# Raw output lands in gpurun_out/<TAG>/; tools/summarize_profiles.py condenses it into the files kept under profiles/.
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 10 --warmup 2"
$BENCH > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
$BENCH --workload c3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
$BENCH --workload c2g --no-cpu-baseline > "$OUT/bench_c2g.json" 2> "$OUT/bench_c2g.err"
$BENCH --workload c3g --no-cpu-baseline > "$OUT/bench_c3g.json" 2> "$OUT/bench_c3g.err"
python $R/bench.py --workload c4 --steps 400 --warmup 100 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python $R/bench.py --workload c5 --steps 400 --warmup 100 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
python $R/bench.py --batch 65536 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_c2_b65536.json" 2> "$OUT/bench_c2_b65536.err"
python $R/bench.py --batch 512 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_c2_b512.json" 2> "$OUT/bench_c2_b512.err"
cd /tmp
NB="--no-cpu-baseline --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c2" -o c2 -- $BENCH $NB > "$OUT/trace_c2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -o c3 -- $BENCH $NB --workload c3 > "$OUT/trace_c3.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c2_b65536" -o b -- python $R/bench.py --batch 65536 --steps 5 --warmup 2 $NB > "$OUT/trace_c2_b65536.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_rf" -o rf -- python $R/tools/gpu_rf_perf.py > "$OUT/trace_rf.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c4" -o c4 -- python $R/bench.py --workload c4 --steps 300 --warmup 100 > "$OUT/trace_c4.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c5" -o c5 -- python $R/bench.py --workload c5 --steps 300 --warmup 100 > "$OUT/trace_c5.log" 2>&1
for WL in c2 c3; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${WL}_$C" -o pmc -- $BENCH $NB --workload $WL --steps 4 --warmup 1 > "$OUT/pmc_${WL}_$C.log" 2>&1
  done
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_${WL}_SQ" -o pmc -- $BENCH $NB --workload $WL --steps 4 --warmup 1 > "$OUT/pmc_${WL}_SQ.log" 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_rf_$C" -o pmc -- python $R/tools/gpu_rf_perf.py > "$OUT/pmc_rf_$C.log" 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_b65536_SQ" -o pmc -- python $R/bench.py --batch 65536 --steps 3 --warmup 1 $NB > "$OUT/pmc_b65536_SQ.log" 2>&1
cd $R
python tools/summarize_profiles.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
