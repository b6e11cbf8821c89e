#!/bin/bash
# Evidence pass of a round on the GPU box: bench lines, per-layer times, rocprofv3 kernel stats and PMC HBM traffic.
#   bash tools/profile_round.sh r03 [stage ...]      stages: pmc bench layers stats   (default: all, pmc first)
# Output under gpurun_out/<round>/; `pmc` also writes profiles/<round>/pmc_traffic.{json,txt} in the box's copy of the tree
# (merged back through gpurun_out/<round>/profiles_<round>/), which bench.py reads for roofline.traffic.
set -e
R=${1:?round name}; shift || true
STAGES=${*:-pmc bench layers stats}
O=gpurun_out/$R; mkdir -p $O
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
PY=$(command -v python3)
for st in $STAGES; do case $st in
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_f32_$c -o pmc -- $PY tools/conv_bench.py --dtype fp32 --layer c52_3x3 --tile 7 --reps 8 > $O/pmc_f32_$c.txt 2>&1
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_f32b_$c -o pmc -- $PY tools/conv_bench.py --dtype fp32 --layer c26_3x3 --tile 6 --reps 8 > $O/pmc_f32b_$c.txt 2>&1
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_wino_$c -o pmc -- $PY tools/conv_bench.py --dtype fp32 --layer c52_3x3 --tile 13 --residual --reps 8 > $O/pmc_wino_$c.txt 2>&1
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_winob_$c -o pmc -- $PY tools/conv_bench.py --dtype fp32 --layer c26_3x3 --tile 13 --residual --reps 8 > $O/pmc_winob_$c.txt 2>&1
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_h16_$c -o pmc -- $PY tools/conv_bench.py --dtype bf16 --layer c104_3x3,c52_3x3,c26_3x3,c13_3x3,c52_1x1 --tile 0 --reps 8 > $O/pmc_h16_$c.txt 2>&1
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_dec_$c -o pmc -- $PY tools/decode_bench.py 0 > $O/pmc_dec_$c.txt 2>&1
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_decwb_$c -o pmc -- $PY tools/decode_bench.py 1 > $O/pmc_decwb_$c.txt 2>&1
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_wg_$c -o pmc -- $PY tools/wgrad_bench.py 3 > $O/pmc_wg_$c.txt 2>&1
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_ws_$c -o pmc -- $PY tools/conv_bench.py --dtype bf16 --layer c208_3x3 --tile 14 --reps 8 > $O/pmc_ws_$c.txt 2>&1
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_sw_$c -o pmc -- $PY tools/wgrad_bench.py stem > $O/pmc_sw_$c.txt 2>&1
  done
  mkdir -p profiles/$R $O/profiles_$R
  $PY tools/pmc_traffic.py profiles/$R \
      conv_patch_f32 conv_patch_f32 692224 $O/pmc_f32_FETCH_SIZE $O/pmc_f32_WRITE_SIZE \
      conv_patch_f32_c26 conv_patch_f32 184320 $O/pmc_f32b_FETCH_SIZE $O/pmc_f32b_WRITE_SIZE \
      conv_wino_f32 conv_wino_f32 - $O/pmc_wino_FETCH_SIZE $O/pmc_wino_WRITE_SIZE \
      wino_xform_f32 wino_xform_f32 - $O/pmc_wino_FETCH_SIZE $O/pmc_wino_WRITE_SIZE \
      conv_wino_f32_c26 conv_wino_f32 - $O/pmc_winob_FETCH_SIZE $O/pmc_winob_WRITE_SIZE \
      wino_xform_f32_c26 wino_xform_f32 - $O/pmc_winob_FETCH_SIZE $O/pmc_winob_WRITE_SIZE \
      conv3_dma_h16 conv3_dma_h16 346112 $O/pmc_h16_FETCH_SIZE $O/pmc_h16_WRITE_SIZE \
      conv3_dma_h16_c104 conv3_dma_h16 692224 $O/pmc_h16_FETCH_SIZE $O/pmc_h16_WRITE_SIZE \
      conv3_dma_h16_c26 conv3_dma_h16 184320 $O/pmc_h16_FETCH_SIZE $O/pmc_h16_WRITE_SIZE \
      conv3_dma_h16_c13 conv3_dma_h16 98304 $O/pmc_h16_FETCH_SIZE $O/pmc_h16_WRITE_SIZE \
      conv1_dma_h16_c52 conv1_dma_h16 - $O/pmc_h16_FETCH_SIZE $O/pmc_h16_WRITE_SIZE \
      decode3 decode3_kernel - $O/pmc_dec_FETCH_SIZE $O/pmc_dec_WRITE_SIZE \
      decode3_write_back decode3_kernel - $O/pmc_decwb_FETCH_SIZE $O/pmc_decwb_WRITE_SIZE \
      wgrad3_dma_h16 wgrad3_dma_h16 - $O/pmc_wg_FETCH_SIZE $O/pmc_wg_WRITE_SIZE \
      wgrad_reduce_acc wgrad_reduce_acc - $O/pmc_wg_FETCH_SIZE $O/pmc_wg_WRITE_SIZE \
      conv3_ws_h16 conv3_ws_h16 - $O/pmc_ws_FETCH_SIZE $O/pmc_ws_WRITE_SIZE \
      stem_wgrad_h16 stem_wgrad_h16 - $O/pmc_sw_FETCH_SIZE $O/pmc_sw_WRITE_SIZE > $O/pmc_traffic_stdout.txt 2>&1 || true
  cp profiles/$R/pmc_traffic.json profiles/$R/pmc_traffic.txt $O/profiles_$R/ 2>/dev/null || true
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o pmc -- $PY tools/conv_bench.py --dtype bf16 --layer c52_3x3,c26_3x3 --tile 0 --reps 8 > $O/pmc_sq.txt 2>&1 && $PY tools/pmc_sum.py $O/pmc_sq conv3 > $O/pmc_sq_summary.txt || echo "SQ pass failed" > $O/pmc_sq_summary.txt
  echo "pmc done";;
bench)
  $PY bench.py > $O/final_bench.json 2> $O/final_bench.log; echo "bench done";;
layers)
  $PY bench.py --per-layer --no-cpu-baseline --train-steps 0 --no-config5 --no-nms > $O/per_layer_bench.json 2> $O/per_layer.txt
  echo "layers done";;
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -o bench -- $PY bench.py --no-cpu-baseline --train-steps 0 > $O/bench_forward_legs_under_rocprof.json 2> $O/bench_forward_legs_under_rocprof.log
  YOLO_TRAIN_TAPE=0 YOLO_WGRAD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_bf16 -o train -- $PY tools/train_bench.py --dtype bf16 --fused-loss --steps 5 > $O/train_bf16.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nms80 -o nms -- $PY tools/nms_bench.py 80 uniform > $O/nms80.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nms2 -o nms -- $PY tools/nms_bench.py 2 uniform > $O/nms2.txt 2>&1
  echo "stats done";;
esac; done
