#!/bin/bash
# gpurun -- bash tools/microbench/run_pab.sh : every built variant of prefill_attn_bench (tools/microbench/pab_*.bin)
cd $GRAFT_REPO_ROOT/tools/microbench
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
for f in pab_*.bin; do echo "== $f"; timeout 120 ./$f; done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/pab.txt
