#!/bin/bash
# tools/r06_check.sh "sections" -- quick checks of a development tree on the GPU box (dev tool; output in gpurun_out/r06chk/)
set -u
SECTIONS=${1:-"tests guard fuzz bench"}
has() { case " $SECTIONS " in *" $1 "*) return 0;; *) return 1;; esac; }
R=$PWD; OUT=$R/gpurun_out/r06chk; mkdir -p "$OUT"; export TMPDIR=/tmp
NB="--no-cpu-baseline --no-parity"
if has tests; then
  timeout 900 python -m pytest tests/test_gpu_swd_lean.py tests/test_gpu_fullsize.py tests/test_gpu_like.py -m gpu -x -q > "$OUT/tests.txt" 2>&1; tail -3 "$OUT/tests.txt"
fi
if has alltests; then
  timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/alltests.txt" 2>&1; tail -3 "$OUT/alltests.txt"
fi
if has chainguard; then
  timeout 600 python tools/gpu_chain_guard.py 8 300 > "$OUT/chain_guard.txt" 2>&1; grep -v amdgpu.ids "$OUT/chain_guard.txt" | tail -4
  timeout 600 python tools/gpu_chain_guard.py 64 200 >> "$OUT/chain_guard.txt" 2>&1; grep -v amdgpu.ids "$OUT/chain_guard.txt" | tail -2
fi
if has guard; then
  timeout 300 python tools/gpu_lean_guard.py > "$OUT/lean_guard.txt" 2>&1; cat "$OUT/lean_guard.txt" | grep -v amdgpu.ids
fi
if has fuzz; then
  LEAN=1 timeout 1200 python tools/gpu_fuzz.py ${FUZZ_SEED:-601} ${FUZZ_N:-2000} > "$OUT/fuzz_lean_${FUZZ_SEED:-601}.txt" 2>&1; tail -4 "$OUT/fuzz_lean_${FUZZ_SEED:-601}.txt"
fi
if has fuzzprior; then
  BH_FUZZ_DUMP=$OUT/fuzz_bad LEAN=1 PRIOR=1 timeout 1200 python tools/gpu_fuzz.py ${FUZZ_SEED:-701} ${FUZZ_N:-2000} > "$OUT/fuzz_lean_prior_${FUZZ_SEED:-701}.txt" 2>&1; tail -4 "$OUT/fuzz_lean_prior_${FUZZ_SEED:-701}.txt"
fi
if has bench; then
  for w in c2 c3; do timeout 300 python bench.py --workload $w $NB --no-rf-roofline > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; python - "$OUT/bench_$w.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['config']['workload'][:40], d['value'], d['ms_per_step'])
PY
  done
  for w in c4 c5 c5_full; do timeout 600 python bench.py --workload $w --steps 600 --warmup 300 > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; python - "$OUT/bench_$w.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], {k:d[k] for k in d if 'guard' in k})
PY
  done
fi
if has trace; then
  cd /tmp
  TR="rocprofv3 --kernel-trace --stats --output-format csv"
  for w in c3 c4; do
    EX=""; [ $w = c4 ] && EX="--steps 300 --warmup 100"; [ $w = c3 ] && EX="--steps 10 --warmup 2"
    timeout 600 $TR -d "$OUT/trace_$w" -o t -- python $R/bench.py --workload $w $EX $NB --no-rf-roofline > "$OUT/trace_$w.log" 2>&1
    f=$(find "$OUT/trace_$w" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4,6-7 "$f" | cut -c1-200 | head -12
  done
  cd $R
fi
