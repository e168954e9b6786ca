# Developer tool (GPU box): does any HIP-runtime knob change what a dependent launch costs?  decode ms per 63 steps (B = 32) of
# tools/decode_probe.py under each setting; results in gpurun_out/runtime_env_sweep.txt
out=gpurun_out/runtime_env_sweep.txt; : > $out
run() { echo "== $*" >> $out; env "$@" timeout 200 python tools/decode_probe.py 32 64 2>&1 | grep decode_ms >> $out; }
run X=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run GPU_FLUSH_ON_EXECUTION=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run AMD_DIRECT_DISPATCH=0
run HIP_FORCE_DEV_KERNARG=1
run ROC_USE_FGS_KERNARG=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run ROC_AQL_QUEUE_SIZE=65536
run MELLOW_NO_GRAPH=1
run X=0
cat $out
