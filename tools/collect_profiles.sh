# Everything under profiles/r06_* in one GPU call (gpurun -- bash tools/collect_profiles.sh).  PMC passes run alone
# (--pmc only, no trace domains); FETCH_SIZE and WRITE_SIZE in separate passes.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r06
O=gpurun_out/$R; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log
# --- PMC: HBM-side traffic of the GEMM family (per launch) and of the decode step (per step) ---
# (counter passes run the prefill as ONE chain: per-kernel counters of two overlapping launches would count each other's cycles,
#  and bench.py's own per-family timing -- which switches the split off like every profiled run -- counts unsplit launches)
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o f --output-format csv -- python tools/pmc_prefill.py --opt prefill_split=1 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o w --output-format csv -- python tools/pmc_prefill.py --opt prefill_split=1 > /dev/null 2>&1
python tools/pmc_traffic.py gemm $O/pf/f_counter_collection.csv $O/pw/w_counter_collection.csv $O/pmc_gemm_traffic.json
# (counter collection costs ~0.1 s per dispatch: the decode passes use a 5-step workload, tools/pmc_decode.py)
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/df -o f --output-format csv -- python tools/pmc_decode.py > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/dw -o w --output-format csv -- python tools/pmc_decode.py > /dev/null 2>&1
python tools/pmc_traffic.py decode $O/df/f_counter_collection.csv $O/dw/w_counter_collection.csv $O/pmc_decode_traffic.json
# (per-kernel busy / waiting / L2-hit shares of the decode kernels: profiles/r03_pmc_decode_counters.txt was collected once with
#  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY and
#  --pmc TCC_HIT_sum TCC_MISS_sum over tools/pmc_decode.py, reduced with tools/pmc_decode_counters.py)
# --- PMC: MFMA utilisation of the GEMM family / attentions ---
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY -d $O/pm -o m --output-format csv -- python tools/pmc_prefill.py --opt prefill_split=1 > $O/pmc_mfma.log 2>&1
python tools/pmc_mfma.py $O/pm/m_counter_collection.csv $O/pmc_gemm_mfma.json || tail -5 $O/pmc_mfma.log
# the same for the fp8 mode (BASELINE configs[4]; B = 32 here like the f32x3 file: per-launch bytes scale with the rows)
MELLOW_PRECISION=fp8 timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pf8 -o f --output-format csv -- python tools/pmc_prefill.py --opt prefill_split=1 > /dev/null 2>&1
MELLOW_PRECISION=fp8 timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pw8 -o w --output-format csv -- python tools/pmc_prefill.py --opt prefill_split=1 > /dev/null 2>&1
MELLOW_PRECISION=fp8 python tools/pmc_traffic.py gemm $O/pf8/f_counter_collection.csv $O/pw8/w_counter_collection.csv $O/pmc_gemm_traffic_fp8.json
cp $O/pmc_gemm_traffic_fp8.json profiles/${R}_pmc_gemm_traffic_fp8.json
rm -rf $O/pf8 $O/pw8
cp $O/pmc_gemm_traffic.json profiles/${R}_pmc_gemm_traffic.json     # bench.py prints `traffic` only from files whose source hash matches
cp $O/pmc_decode_traffic.json profiles/${R}_pmc_decode_traffic.json
# --- bench lines ---
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 5 --warmup 2 --precision f32 --no-cpu-baseline --no-alt-modes --no-b64 --no-configs3 --inflight 0 > $O/bench_f32.json 2>/dev/null
timeout 300 python bench.py --steps 5 --warmup 2 --precision fp8 --no-cpu-baseline > $O/bench_fp8.json 2>/dev/null
timeout 300 python bench.py --steps 3 --warmup 1 --preset configs2 --no-cpu-baseline --no-alt-modes --no-b64 --no-configs3 --inflight 0 > $O/bench_configs2.json 2>/dev/null
timeout 300 python bench.py --steps 3 --warmup 1 --preset configs4 --no-cpu-baseline --no-alt-modes --no-b64 --no-configs3 --inflight 0 > $O/bench_configs4.json 2>/dev/null
timeout 300 python bench.py --steps 3 --warmup 1 --preset configs3 --no-cpu-baseline > $O/bench_configs3.json 2>/dev/null
# --- kernel trace + stats of the default command, decode timeline ---
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-modes --no-b64 --no-configs3 --inflight 0 > $O/bench_under_rocprofv3.json 2>/dev/null
cp $O/stats/st_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null || find $O/stats -name "*stats*.csv" | head
# the same with the prefill as ONE chain (--option prefill_split=1): kernel durations do not overlap, so the GEMM family time of
# `roofline_gemm` can be re-derived from this file alone
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats1 -o st --output-format csv -- python bench.py --option prefill_split=1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-modes --no-b64 --no-configs3 --inflight 0 > $O/bench_onechain_under_rocprofv3.json 2>/dev/null
cp $O/stats1/st_kernel_stats.csv $O/bench_onechain_kernel_stats.csv 2>/dev/null
timeout 300 rocprofv3 --kernel-trace -d $O/trace_dec -o tr --output-format csv -- python tools/decode_probe.py > /dev/null 2>&1
python tools/trace_summary.py $O/trace_dec/tr_kernel_trace.csv 40 > $O/decode_step_timeline.txt 2>&1
timeout 400 python tools/fp8_agreement.py structured 2>&1 | grep -v amdgpu.ids > $O/fp8_agreement.txt
timeout 400 python tools/fp8_agreement.py structured --opt fp8_kv16=0 2>&1 | grep -v amdgpu.ids > $O/fp8_agreement_fp32_pages.txt
# --- decode phase by batch size: the f32x3 multi-row-block path against the fp32 kernels replicated per row block ---
{ echo "# decode phase of one generate() pass, 63 steps, f32x3 mode (tools/decode_probe.py); x3 = f32x3 forms of the layer GEMM launches from two row blocks on + the streaming lm_head; fp32 = --opt decode_x3=0 (round 4's kernels)";
  for B in 32 64 128 256 512; do echo "B=$B x3  : $(timeout 300 python tools/decode_probe.py $B 64 2>&1 | grep decode_ms)"; echo "B=$B fp32: $(timeout 300 python tools/decode_probe.py $B 64 --opt decode_x3=0 2>&1 | grep decode_ms)"; done; } > $O/decode_batch_table.txt
timeout 300 rocprofv3 --kernel-trace -d $O/trace_dec64 -o tr --output-format csv -- python tools/decode_probe.py 64 64 > /dev/null 2>&1
python tools/trace_summary.py $O/trace_dec64/tr_kernel_trace.csv 40 > $O/decode_step_timeline_b64.txt 2>&1
rm -rf $O/trace_dec64
rm -rf $O/stats $O/stats1 $O/pf $O/pw $O/df $O/dw $O/pm $O/trace_dec
ls -la $O
