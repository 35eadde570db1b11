python3 -m pytest tests/test_gpu_devbfgs.py tests/test_gpu_fast.py -x -q -m gpu > gpurun_out/t3.log 2>&1; tail -2 gpurun_out/t3.log
bash tools/ab_libs.sh "libnghmm_base.so libnghmm_new.so" "c2 c2r" 40 2>&1
