#!/bin/bash
# usage: [PMC_SCRIPT=fineq_lab.py] tools/lab/pmc.sh <tag> <kernel substring> <script args...>   (run on the GPU box; the script defaults to one_xw.py)
tag=$1; pat=$2; shift; shift
root=$(pwd); out=$root/gpurun_out/lab/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o p -- python3 $root/tools/lab/${PMC_SCRIPT:-one_xw.py} "$@" > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; }
done
cd $root
python3 tools/lab/pmc_sum.py $out "$pat" > $out/summary.txt; cat $out/summary.txt
rm -rf $out/p[0-9]
