#!/bin/bash
# Round-3 batch 4: SQ / SQC counters of the one-ray and the two-rays-per-lane chain bodies on relay4 (fused read-out and
# trace only) and on the 8-element C4 chain.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
G="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM|SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR"
tools/box_state.sh gpurun_out/r03_exp4_box.txt
for cfg in relay4 C4; do
  for rpl in 1 2; do
    export ART_CHAIN_RPL=$rpl
    if [ $rpl = 2 ]; then export ART_CHAIN_WAVES=4; else unset ART_CHAIN_WAVES; fi
    timeout -k 10 500 bash tools/prof_counters.sh ${cfg}_rpl$rpl "$G" --config $cfg --steps 10 --warmup 3 || { echo "failed $cfg $rpl"; tail -3 gpurun_out/cnt_${cfg}_rpl$rpl/*.err; exit 1; }
    echo "== $cfg rpl$rpl"; python3 tools/summarize_counters.py gpurun_out/cnt_${cfg}_rpl$rpl
  done
done
