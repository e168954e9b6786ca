# usage: bash tools/trace_decode.sh <tag> [B] [extra env...]   -> gpurun_out/trace_<tag>.txt (per-kernel averages of the decode probe)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; B=${2:-32}
O=gpurun_out/tr_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O -o tr --output-format csv -- python tools/decode_probe.py $B 64 > $O/probe.log 2>&1
python tools/trace_summary.py $O/tr_kernel_trace.csv 14 > gpurun_out/trace_$tag.txt 2>&1
tail -2 $O/probe.log >> gpurun_out/trace_$tag.txt
rm -rf $O
cat gpurun_out/trace_$tag.txt
