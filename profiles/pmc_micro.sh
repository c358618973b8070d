#!/bin/bash
# usage (GPU box, repo root): bash profiles/pmc_micro.sh TAG "<conv_micro.py args>"  -> gpurun_out/pmc_TAG_*.csv (several counter passes)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -d $R/gpurun_out/pmc_${TAG}_$i -o p --output-format csv -- python3 $R/profiles/conv_micro.py "$@" > /dev/null 2>&1
done
cd $R && python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in sorted(glob.glob("gpurun_out/pmc_${TAG}_*/p_counter_collection.csv")):
    seen=set()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    if "conv_p" in k or "wgrad" in k:
        print(k)
        for c,x in sorted(v.items()): print("   %-34s %.4g" % (c,x))
PY
