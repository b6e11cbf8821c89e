#!/bin/bash
# Evidence pass of a round on the GPU box: bench lines, per-layer times, rocprofv3 kernel stats and PMC HBM traffic.
#   bash tools/profile_round.sh r02 [stage ...]      stages: bench layers stats pmc   (default: all)
# Output under gpurun_out/<round>/; the files to keep are copied into profiles/<round>/ by hand afterwards.
set -e
R=${1:?round name}; shift || true
STAGES=${*:-bench layers stats pmc}
O=gpurun_out/$R; mkdir -p $O
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
PY=$(command -v python3)
for st in $STAGES; do case $st in
bench)
  $PY bench.py > $O/final_bench.json 2> $O/final_bench.log; echo "bench done";;
layers)
  $PY bench.py --per-layer --no-cpu-baseline --train-steps 0 --no-config5 --no-nms > $O/per_layer_bench.json 2> $O/per_layer.txt
  echo "layers done";;
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- $PY bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_bf16 -o train -- $PY tools/train_bench.py --dtype bf16 --fused-loss --steps 5 > $O/train_bf16.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nms80 -o nms -- $PY tools/nms_bench.py 80 uniform > $O/nms80.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nms2 -o nms -- $PY tools/nms_bench.py 2 uniform > $O/nms2.txt 2>&1
  echo "stats done";;
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- $PY tools/conv_bench.py --dtype bf16 --layer c104_3x3,c52_3x3,c26_3x3,c13_3x3 --tile 0 --reps 8 > $O/pmc_$c.txt 2>&1
    $PY tools/pmc_sum.py $O/pmc_$c conv >> $O/pmc_summary.txt
  done
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o pmc -- $PY tools/conv_bench.py --dtype bf16 --layer c52_3x3,c26_3x3 --tile 0 --reps 8 > $O/pmc_sq.txt 2>&1 && $PY tools/pmc_sum.py $O/pmc_sq conv3 >> $O/pmc_summary.txt || echo "SQ pass failed" >> $O/pmc_summary.txt
  echo "pmc done";;
esac; done
