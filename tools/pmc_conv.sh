#!/bin/bash
# Dev tool: SQ counters for a few conv shapes (run on the GPU box):  tools/pmc_conv.sh <tag> "16,16,128,3 128,128,16,3 ..."
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT \
  -d $out/p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_conv_one.py $@ > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS \
  -d $out/p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_conv_one.py $@ > $out/p2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 \
  -d $out/p3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_conv_one.py $@ > $out/p3.log 2>&1 || true
find $out -name "*counter_collection.csv" | head
