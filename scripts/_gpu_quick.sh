# scratch helper (GPU box): the GPU test suite + one bench line
mkdir -p gpurun_out/quick
timeout 2400 python -m pytest tests -m gpu -q "$@" > gpurun_out/quick/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/quick/gpu_tests.txt
tail -12 gpurun_out/quick/gpu_tests.txt
