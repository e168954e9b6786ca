set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python tools/pmc_prefill.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python tools/pmc_prefill.py > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $O/pmc_gemm_traffic.json
cp $O/pmc_gemm_traffic.json profiles/r02_pmc_gemm_traffic.json   # bench.py prints `traffic` only from a file whose source hash matches
timeout 400 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
timeout 200 python bench.py --steps 5 --warmup 2 --precision f32 --no-cpu-baseline --no-alt-modes --no-b64 > $O/bench_f32.json 2>/dev/null
timeout 200 python bench.py --steps 5 --warmup 2 --precision fp8 --no-cpu-baseline > $O/bench_fp8.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-modes --no-b64 > $O/bench_under_rocprofv3.json 2>/dev/null
cp $O/stats/st_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null || find $O/stats -name "*stats*.csv" | head
rocprofv3 --kernel-trace -d $O/trace_dec -o tr --output-format csv -- python tools/decode_probe.py > /dev/null 2>&1
python tools/trace_summary.py $O/trace_dec/tr_kernel_trace.csv 40 > $O/decode_step_timeline.txt 2>&1
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/trace_dec
ls -la $O
