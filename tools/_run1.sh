mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputest_b.log 2>&1; tail -8 gpurun_out/r05_gputest_b.log
