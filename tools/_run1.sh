cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32x3 and (lm_prefill_and_decode or generate_tokens or batch32_matches or encoder_taps or prefix or norm_free or split_prefill)" 2>&1 | tail -3
for B in 32 64; do echo "ngroup on : $(timeout 300 python tools/decode_probe.py $B 64 2>&1 | grep decode_ms | cut -c100-260)"; echo "ngroup off: $(MELLOW_X3Q_NGROUP=0 timeout 300 python tools/decode_probe.py $B 64 2>&1 | grep decode_ms | cut -c100-260)"; done
echo "one chain on : $(MELLOW_PREFILL_SPLIT=1 timeout 300 python tools/decode_probe.py 32 64 2>&1 | grep decode_ms | cut -c100-260)"
echo "one chain off: $(MELLOW_X3Q_NGROUP=0 MELLOW_PREFILL_SPLIT=1 timeout 300 python tools/decode_probe.py 32 64 2>&1 | grep decode_ms | cut -c100-260)"
export MELLOW_PREFILL_SPLIT=1
O=gpurun_out/shape; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o f --output-format csv -- python tools/pmc_prefill.py > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o w --output-format csv -- python tools/pmc_prefill.py > /dev/null 2>&1
python tools/pmc_by_shape.py $O/pf/f_counter_collection.csv $O/pw/w_counter_collection.csv > gpurun_out/r05_pmc_traffic_by_shape.txt 2>&1
rm -rf $O
head -6 gpurun_out/r05_pmc_traffic_by_shape.txt
