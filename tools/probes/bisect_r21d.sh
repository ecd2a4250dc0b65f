mkdir -p gpurun_out/r4
run() { tag=$1; shift; env "$@" timeout -k 10 300 python -m pytest tests/test_models_gpu.py -q -m gpu -x -s -k "test_train_steps_fp32_against_reference_fixture and simclr_timeseriesv4-r21d" > gpurun_out/r4/bis_$tag.log 2>&1; echo "$tag $* -> rc $? $(grep -E "^E  +(Assertion|assert)" gpurun_out/r4/bis_$tag.log | head -2 | tr '\n' ' ')"; }
run a DUALVAR_CONV_TAP_BM128=0
run b DUALVAR_CONV_TAP_GRID=128
run c DUALVAR_CONV_TAP_BM128=0 DUALVAR_CONV_TAP_GRID=128
run d DUALVAR_CONV_TAP_BM128=0 DUALVAR_CONV_PP_FWD=0
run e DUALVAR_CONV_TAP_GRID=128 DUALVAR_CONV_PP_FWD=0
