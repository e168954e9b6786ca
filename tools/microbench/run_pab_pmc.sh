#!/bin/bash
# gpurun -- bash tools/microbench/run_pab_pmc.sh : SQ counters of the prefill attention microbenchmark (pab_base.bin)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pab_pmc; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/a -o a --output-format csv -- tools/microbench/pab_base.bin > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU -d $O/b -o b --output-format csv -- tools/microbench/pab_base.bin > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pab_pmc/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:40], r["Grid_Size"]) if "Grid_Size" in r else (r["Kernel_Name"][:40], "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
