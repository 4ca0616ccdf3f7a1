#!/bin/bash
# the model-level parity tests under other arithmetic orderings of the same build -- Block.norm2 inside the fused-MLP prologue instead of the
# proj epilogue (MVLT_NO_PROJ_LN=1), the sigmoid-form GELU instead of the polynomials (a second library built with -DMVLT_GELU_POLY=0), 128-wide GEMMs instead of
# the 8-phase ones (MVLT_NT_P8=0): the tightest margins of each run.  The sigmoid build: bash tools/build_alt.sh sigm gemm.hip,mlp.hip -DMVLT_GELU_POLY=0   gpurun -- 'bash tools/parity_orderings.sh > gpurun_out/rNN_parity_orderings.txt'
cd "${GRAFT_REPO_ROOT:-.}"
run() {
  echo "=== $1"
  env $2 python -m pytest tests/test_model_gpu.py tests/test_engine_gpu.py -m gpu -q 2>&1 | tail -1
  python tools/parity_report.py "$1" 2>/dev/null | sed -n '/# tightest margins/,$p' | head -7
}
run "default build" "MVLT_DUMMY=1"
run "MVLT_NO_PROJ_LN=1" "MVLT_NO_PROJ_LN=1"
run "MVLT_NT_P8=0 MVLT_NO_LIN_FUSE=1 (128-wide GEMMs instead of the 8-phase ones, weight and input gradients as two launches)" "MVLT_NT_P8=0 MVLT_NO_LIN_FUSE=1"
run "MVLT_MIM_FP32_Z=1 MVLT_NO_OUT_OP=1 (fp32 pre-BatchNorm conv outputs, fp32 stage outputs + cast pass)" "MVLT_MIM_FP32_Z=1 MVLT_NO_OUT_OP=1"
run "MVLT_TN_P8=0 (weight-gradient split reductions by fp32 atomics instead of bf16 partial tiles + fold)" "MVLT_TN_P8=0"
run "MVLT_TN_DEFER_FOLD=0 (one fold launch behind every weight-gradient GEMM instead of the batched folds)" "MVLT_TN_DEFER_FOLD=0"
run "MVLT_TN_P8_320=0 MVLT_TN_NO_OVERWRITE=1 MVLT_MLP_DW_PARTIALS=0 MVLT_TN_PART_MINOUT=65536 (round 6's reductions off: the stage-3 fc gradients on the 128-wide kernel, the vocabulary / fused-MLP / small-output gradients by atomics)" "MVLT_TN_P8_320=0 MVLT_TN_NO_OVERWRITE=1 MVLT_MLP_DW_PARTIALS=0 MVLT_TN_PART_MINOUT=65536"
[ -f ab/libmvlt_sigm.so ] && run "sigmoid-form GELU build (ab/libmvlt_sigm.so)" "MVLT_HIP_LIB=ab/libmvlt_sigm.so"
[ -f ab/libmvlt_sigm.so ] && run "sigmoid-form GELU build + MVLT_NO_PROJ_LN=1" "MVLT_HIP_LIB=ab/libmvlt_sigm.so MVLT_NO_PROJ_LN=1"
