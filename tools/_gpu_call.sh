python -m pytest tests -m gpu -q > gpurun_out/r2_t16.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2_t16.log
