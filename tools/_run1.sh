mkdir -p gpurun_out; rm -f gpurun_out/trace_*.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32x3 and (lm_prefill_and_decode or generate_tokens or batch32_matches or batch64_matches or multi_row_block or row_block_early_exit or late_positions)" > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
bash tools/trace_decode.sh d3_b32 32 > /dev/null 2>&1
for v in d1 d2 nw4 nw6; do MELLOW_HIP_LIB=mellow_amd/lib/ab/libmellow_hip_$v.so bash tools/trace_decode.sh ${v}_b32 32 > /dev/null 2>&1; done
bash tools/trace_decode.sh d3_b64 64 > /dev/null 2>&1
for f in gpurun_out/trace_*.txt; do echo "== $f"; grep -E "fullk|head3|decode_ms" $f; done
