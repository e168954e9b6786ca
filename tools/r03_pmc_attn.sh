cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/pmc_attn; rm -rf $O
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $O -o a --output-format csv -- python tools/pmc_prefill.py > $O.log 2>&1
python tools/pmc_kernel_avg.py $O/a_counter_collection.csv prefill_attention gemm_x3q_kernel > gpurun_out/pmc_attn.txt 2>&1
rm -rf $O
cat gpurun_out/pmc_attn.txt
