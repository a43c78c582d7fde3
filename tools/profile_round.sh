#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 evidence of one round, run ON THE GPU BOX:
#   gpurun --timeout 3000 -- "BH_PROFILE_COMMIT=$(git rev-parse --short HEAD) bash tools/profile_round.sh r04"
# Bench lines; kernel-trace statistics of c2 / c3 / the chain workloads / the receiver function alone (ONE batch
# shape per pass) / the Gauss-law contraction / the B = 65536 throughput regime; and, in SEPARATE passes (never
# together with a trace domain), the HBM counters (FETCH_SIZE / WRITE_SIZE) and the SQ activity counters of the c2, c3
# and RF-alone (c3 shape) commands, plus the c2 HBM passes with the progress board off (attribution of its traffic).
# (Under --pmc the engine runs without its start gate between the dispersion kernel and the RF stream -- counter
# collection serialises the dispatches of all queues, see bh_engine_create -- so the c3 counter passes show the RF kernels'
# own figures, not their placement beside the dispersion kernel.)
# Raw output lands in gpurun_out/<TAG>/; tools/summarize_profiles.py condenses it into the files kept under profiles/.
# A second argument selects sections (default "bench trace pmc"): e.g. `bash tools/profile_round.sh r03 "bench trace"`.
set -u
TAG=${1:-r03}
SECTIONS=${2:-"bench trace pmc"}
has() { case " $SECTIONS " in *" $1 "*) return 0;; *) return 1;; esac; }
stamp() { echo "[profile_round] $(date +%T) $*"; }
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
NB="--no-cpu-baseline --no-parity"
if has bench; then
stamp "bench lines"
python $R/bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python $R/bench.py --workload c2g --no-cpu-baseline > "$OUT/bench_c2g.json" 2> "$OUT/bench_c2g.err"
python $R/bench.py --workload c3g --no-cpu-baseline > "$OUT/bench_c3g.json" 2> "$OUT/bench_c3g.err"
python $R/bench.py --workload c2 --batch 65536 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_c2_b65536.json" 2> "$OUT/bench_c2_b65536.err"
python $R/bench.py --workload c2 --batch 65536 --steps 5 --warmup 2 --no-cpu-baseline --search reference > "$OUT/bench_c2_b65536_reference.json" 2> "$OUT/bench_c2_b65536_reference.err"
python $R/bench.py --workload c2 --batch 512 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_c2_b512.json" 2> "$OUT/bench_c2_b512.err"
python $R/bench.py --workload c2 --batch 16384 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_c2_b16384.json" 2> "$OUT/bench_c2_b16384.err"
# the short refinement with the reference's arithmetic (the layer-parallel kernel); the chains with the fast arithmetic
python $R/bench.py --workload c2 --arith exact --no-cpu-baseline > "$OUT/bench_c2_fastexact.json" 2> "$OUT/bench_c2_fastexact.err"
for w in c4 c5 c5_full; do python $R/bench.py --workload $w --steps 600 --warmup 300 --arith fast > "$OUT/bench_${w}_arithfast.json" 2> "$OUT/bench_${w}_arithfast.err"; done
# the trial-per-lane kernel: rounds and cycles per round; the fast arithmetic's cost and distance from the exact one
python $R/tools/gpu_lean_rounds.py > "$OUT/lean_rounds.txt" 2>&1
python $R/tools/gpu_lean_guard.py > "$OUT/lean_guard.txt" 2>&1
# the guard in a sampler's windows (8 and 64 chains): how many windows hold a guarded model, and by which rule
{ python $R/tools/gpu_chain_guard.py 8 300; python $R/tools/gpu_chain_guard.py 64 200; } 2>&1 | grep -v amdgpu.ids > "$OUT/chain_guard.txt"
# c2's targets on models drawn from the chains' prior (ragged 2..21 layers, velocities in any order)
python $R/bench.py --workload c2p --no-cpu-baseline > "$OUT/bench_c2p.json" 2> "$OUT/bench_c2p.err"
[ -x $R/tools/ubench/faeval ] && $R/tools/ubench/faeval > "$OUT/faeval.txt" 2>&1
[ -x $R/tools/ubench/fadiff ] && $R/tools/ubench/fadiff > "$OUT/fadiff.txt" 2>&1
python $R/bench.py --workload c4 --steps 700 --warmup 300 --spec-depth 1 > "$OUT/bench_c4_depth1.json" 2> "$OUT/bench_c4_depth1.err"
python $R/bench.py --workload c5 --steps 700 --warmup 300 --spec-depth 1 > "$OUT/bench_c5_depth1.json" 2> "$OUT/bench_c5_depth1.err"
BH_SWD_SEARCH=reference python $R/tools/gpu_latency.py > "$OUT/latency.txt" 2>&1
# the reference's own sequence (bh_engine_set_swd_search): c2 line; chains with every search mode; single-model latency with the default search
python $R/bench.py --workload c2 --search reference --no-cpu-baseline > "$OUT/bench_c2_reference.json" 2> "$OUT/bench_c2_reference.err"
for w in c4 c5; do for sm in reference fast_rayleigh fast; do
  python $R/bench.py --workload $w --steps 600 --warmup 300 --search $sm > "$OUT/bench_${w}_$sm.json" 2> "$OUT/bench_${w}_$sm.err"
done; done
python $R/tools/gpu_latency.py > "$OUT/latency_fast.txt" 2>&1
# the counted Love scan (bh_engine_set_swd_scan): a launch of Love targets only, B = 4096, every step / counted; B = 65536
export BH_SWD_SEARCH=reference
{ echo "== Love only, B = 4096, scan auto (counted)"; TARGETS=L python $R/tools/gpu_trace.py 0 0 | head -6
  echo "== Love only, B = 4096, scan steps"; BH_SWD_SCAN=steps TARGETS=L python $R/tools/gpu_trace.py 0 0 | head -6
  echo "== Rayleigh + Love, B = 4096, scan counted everywhere"; BH_SWD_SCAN=counted python $R/tools/gpu_trace.py 0 0 | head -7
  echo "== Rayleigh + Love, B = 4096, scan auto (= steps here)"; python $R/tools/gpu_trace.py 0 0 | head -7
  for sc in auto steps; do for sm in reference fast; do
    echo "== c2 at B = 65536, scan $sc, search $sm: $(BH_SWD_SCAN=$sc python $R/bench.py --workload c2 --batch 65536 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --search $sm 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', round(d['value']), 'evals/s')")"
  done; done; } > "$OUT/love_scan.txt" 2>&1
# the dispersion kernel's phase clocks with and without the receiver function beside it (c3 against c2)
python $R/tools/gpu_phase_c3.py > "$OUT/phase_c3.txt" 2>&1
python $R/tools/gpu_c3_tail.py > "$OUT/c3_tail.txt" 2>&1
unset BH_SWD_SEARCH
BH_SWD_ARITH=exact python $R/tools/gpu_phase_c3.py > "$OUT/phase_c3_fastexact.txt" 2>&1
# randomised parity sweeps (tools/gpu_fuzz.py): reference sequence; short refinement with its guard (BH_FUZZ_REF=0 BH_FUZZ_FAST=0: skip)
[ "${BH_FUZZ_REF:-3000}" != 0 ] && python $R/tools/gpu_fuzz.py 404 ${BH_FUZZ_REF:-3000} > "$OUT/fuzz_reference.txt" 2>&1
[ "${BH_FUZZ_FAST:-8000}" != 0 ] && FAST=1 python $R/tools/gpu_fuzz.py 405 ${BH_FUZZ_FAST:-8000} > "$OUT/fuzz_fast.txt" 2>&1
[ "${BH_FUZZ_LEAN:-8000}" != 0 ] && LEAN=1 python $R/tools/gpu_fuzz.py 406 ${BH_FUZZ_LEAN:-8000} > "$OUT/fuzz_lean.txt" 2>&1
[ "${BH_FUZZ_LEAN:-8000}" != 0 ] && LEAN=1 PRIOR=1 python $R/tools/gpu_fuzz.py 407 ${BH_FUZZ_LEAN:-8000} > "$OUT/fuzz_lean_prior.txt" 2>&1   # models drawn from a sampler's prior
for sh in c3 tut t512u t512r n8192 n16384; do python $R/tools/gpu_rf_perf.py $sh 2>&1 | tail -1; done > "$OUT/rf_alone.txt"
for s in "4096 1024" "4096 2048" "8192 1024" "1024 1024" "4096 201"; do python $R/tools/gpu_gauss_perf.py $s 2>&1 | tail -1; done > "$OUT/gauss_alone.txt"
fi
cd /tmp
TR="rocprofv3 --kernel-trace --stats --output-format csv"
if has trace; then
stamp "kernel traces"
$TR -d "$OUT/trace_c2" -o t -- python $R/bench.py --workload c2 --search reference --steps 10 --warmup 2 $NB > "$OUT/trace_c2.log" 2>&1
$TR -d "$OUT/trace_c3" -o t -- python $R/bench.py --workload c3 --search reference --steps 10 --warmup 2 $NB --no-rf-roofline > "$OUT/trace_c3.log" 2>&1
$TR -d "$OUT/trace_c3fast" -o t -- python $R/bench.py --workload c3 --search fast --steps 10 --warmup 2 $NB --no-rf-roofline > "$OUT/trace_c3fast.log" 2>&1
$TR -d "$OUT/trace_c3g" -o t -- python $R/bench.py --workload c3g --steps 10 --warmup 2 $NB --no-rf-roofline > "$OUT/trace_c3g.log" 2>&1
$TR -d "$OUT/trace_c2_b65536" -o t -- python $R/bench.py --workload c2 --search reference --batch 65536 --steps 5 --warmup 2 $NB > "$OUT/trace_c2_b65536.log" 2>&1
$TR -d "$OUT/trace_c2fast" -o t -- python $R/bench.py --workload c2 --search fast --steps 10 --warmup 2 $NB > "$OUT/trace_c2fast.log" 2>&1
$TR -d "$OUT/trace_c2fastexact" -o t -- python $R/bench.py --workload c2 --search fast --arith exact --steps 10 --warmup 2 $NB > "$OUT/trace_c2fastexact.log" 2>&1
$TR -d "$OUT/trace_c4" -o t -- python $R/bench.py --workload c4 --steps 700 --warmup 300 > "$OUT/trace_c4.log" 2>&1
$TR -d "$OUT/trace_c5" -o t -- python $R/bench.py --workload c5 --steps 600 --warmup 300 > "$OUT/trace_c5.log" 2>&1
for sh in c3 tut t512u t512r n16384; do
  $TR -d "$OUT/trace_rf_$sh" -o t -- python $R/tools/gpu_rf_perf.py $sh 20 > "$OUT/trace_rf_$sh.log" 2>&1
done
$TR -d "$OUT/trace_gauss" -o t -- python $R/tools/gpu_gauss_perf.py 4096 1024 20 > "$OUT/trace_gauss.log" 2>&1
fi
if has tracefast && ! has trace; then   # (only the traces of the default settings, c2 and c3)
stamp "kernel traces (default settings)"
$TR -d "$OUT/trace_c3fast" -o t -- python $R/bench.py --workload c3 --search fast --steps 10 --warmup 2 $NB --no-rf-roofline > "$OUT/trace_c3fast.log" 2>&1
$TR -d "$OUT/trace_c2fast" -o t -- python $R/bench.py --workload c2 --search fast --steps 10 --warmup 2 $NB > "$OUT/trace_c2fast.log" 2>&1
fi
if has pmc || has pmcfast; then   # (pmcfast: only the passes of the default settings, c2 and c3)
stamp "counter passes"
SMS="fast reference"; has pmc || SMS="fast"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"
SQ2="SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA"
PM="rocprofv3 --output-format csv"
for WL in c2 c3; do for SM in $SMS; do      # (keys: c2fast / c3fast = the engine's default search, c2 / c3 = the reference's sequence)
  KEY=$WL; [ $SM = fast ] && KEY=${WL}fast
  CMD="python $R/bench.py --workload $WL --search $SM --steps 4 --warmup 1 $NB --no-rf-roofline"
  for C in FETCH_SIZE WRITE_SIZE; do
    $PM --pmc $C -d "$OUT/pmc_${KEY}_$C" -o pmc -- $CMD > "$OUT/pmc_${KEY}_$C.log" 2>&1
  done
  $PM --pmc $SQ -d "$OUT/pmc_${KEY}_SQ" -o pmc -- $CMD > "$OUT/pmc_${KEY}_SQ.log" 2>&1
  $PM --pmc $SQ2 -d "$OUT/pmc_${KEY}_SQ2" -o pmc -- $CMD > "$OUT/pmc_${KEY}_SQ2.log" 2>&1
done; done
fi
if has pmc; then
CMD="python $R/bench.py --workload c2 --search fast --arith exact --steps 4 --warmup 1 $NB"   # the short refinement in the reference's arithmetic
for C in FETCH_SIZE WRITE_SIZE; do $PM --pmc $C -d "$OUT/pmc_c2fastexact_$C" -o pmc -- $CMD > "$OUT/pmc_c2fastexact_$C.log" 2>&1; done
$PM --pmc $SQ -d "$OUT/pmc_c2fastexact_SQ" -o pmc -- $CMD > "$OUT/pmc_c2fastexact_SQ.log" 2>&1
$PM --pmc $SQ2 -d "$OUT/pmc_c2fastexact_SQ2" -o pmc -- $CMD > "$OUT/pmc_c2fastexact_SQ2.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do   # the progress board's share of the c2 traffic: the same passes with the board off
  BH_SWD_NO_BOARD=1 $PM --pmc $C -d "$OUT/pmc_c2noboard_$C" -o pmc -- python $R/bench.py --workload c2 --search reference --steps 4 --warmup 1 $NB > "$OUT/pmc_c2noboard_$C.log" 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  $PM --pmc $C -d "$OUT/pmc_rf_c3_$C" -o pmc -- python $R/tools/gpu_rf_perf.py c3 5 > "$OUT/pmc_rf_c3_$C.log" 2>&1
done
$PM --pmc $SQ -d "$OUT/pmc_rf_c3_SQ" -o pmc -- python $R/tools/gpu_rf_perf.py c3 5 > "$OUT/pmc_rf_c3_SQ.log" 2>&1
$PM --pmc $SQ2 -d "$OUT/pmc_rf_c3_SQ2" -o pmc -- python $R/tools/gpu_rf_perf.py c3 5 > "$OUT/pmc_rf_c3_SQ2.log" 2>&1
$PM --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d "$OUT/pmc_gauss_SQ" -o pmc -- python $R/tools/gpu_gauss_perf.py 4096 1024 5 > "$OUT/pmc_gauss_SQ.log" 2>&1
$PM --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d "$OUT/pmc_b65536_SQ" -o pmc -- python $R/bench.py --workload c2 --search reference --batch 65536 --steps 3 --warmup 1 $NB > "$OUT/pmc_b65536_SQ.log" 2>&1
fi
stamp "summaries"
cd $R
python tools/summarize_profiles.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
# the kernels of one steady step of the fused call and of a sampler's window, from the kernel traces (who runs beside whom)
[ -d "$OUT/trace_c3fast" ] && python tools/trace_timeline.py "$OUT/trace_c3fast" > "profiles/${TAG}_c3_timeline.txt" 2>&1
[ -d "$OUT/trace_c4" ] && python tools/trace_timeline.py "$OUT/trace_c4" > "profiles/${TAG}_c4_timeline.txt" 2>&1
cat "$OUT/summary.txt"
