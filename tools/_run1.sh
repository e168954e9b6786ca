mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputest_a.log 2>&1; tail -15 gpurun_out/r05_gputest_a.log
