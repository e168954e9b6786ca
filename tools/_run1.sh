timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32x3 and (lm_prefill_and_decode or generate_tokens or batch32_matches or all_position or late_positions or config4 or split_prefill or norm_free or long_audio or edge_shapes or ragged)" 2>&1 | tail -3
for B in 32 64; do echo "pf2: $(timeout 300 python tools/decode_probe.py $B 64 2>&1 | grep decode_ms | cut -c100-260)"; done
bash tools/trace_decode.sh pf2 32 > /dev/null 2>&1; grep -E "prefill_attention" gpurun_out/trace_pf2.txt
