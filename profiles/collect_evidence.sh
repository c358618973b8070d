#!/bin/bash
# usage: bash profiles/collect_evidence.sh MODE [STORAGE] [TAG] [WORKLOAD]   (on the GPU box, from the repo root)
#   MODE = --precision of bench.py, STORAGE = --storage (default f32), WORKLOAD = --workload (default full3),
#   TAG = file prefix (default r04_${MODE}[_s16][_${WORKLOAD}]); writes gpurun_out/${TAG}_*: copy those into profiles/
MODE=$1
STORAGE=${2:-f32}
WORKLOAD=${4:-full3}
if [ "$STORAGE" = "f32" ]; then DEF=r04_${MODE}; else DEF=r04_${MODE}_s16; fi
if [ "$WORKLOAD" != "full3" ]; then DEF=${DEF}_${WORKLOAD}; fi
TAG=${3:-$DEF}
R=$GRAFT_REPO_ROOT
case $MODE in f16*) LOWC=SQ_INSTS_VALU_MFMA_MOPS_F16;; *) LOWC=SQ_INSTS_VALU_MFMA_MOPS_BF16;; esac
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --graph off --workload $WORKLOAD --precision $MODE --storage $STORAGE"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ev_${TAG}_stats -o p -- $B > /dev/null 2>&1 &&
# (the counter passes run the three discriminator updates on ONE stream: a kernel's counters are per dispatch, and kernels of different
#  streams sharing the chip dilute each other's busy fractions; the kernel-time shares come from the pass above, which overlaps them)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 $LOWC --kernel-trace --output-format csv -d $R/gpurun_out/ev_${TAG}_mfma -o p -- $B --single-stream > /dev/null 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ev_${TAG}_fetch -o p -- $B --single-stream > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ev_${TAG}_write -o p -- $B --single-stream > /dev/null 2>&1 &&
# (instruction mix: on gfx950 the fp32 MFMA shares the vector lanes with the VALU -- DESIGN.md §4 "What an fp32 MFMA kernel pays for" -- so the
#  non-MFMA VALU instructions per MFMA instruction of a family say how far it can get; optional: a failure here does not stop the summary)
{ rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/ev_${TAG}_insts -o p -- $B --single-stream > /dev/null 2>&1 || rm -rf $R/gpurun_out/ev_${TAG}_insts; } &&
cd $R && python3 profiles/make_counters.py gpurun_out/ev_${TAG} gpurun_out ${TAG} > gpurun_out/ev_${TAG}_summary.txt 2>&1
tail -3 gpurun_out/ev_${TAG}_summary.txt
