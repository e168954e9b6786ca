mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputest_d.log 2>&1; tail -4 gpurun_out/r05_gputest_d.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
