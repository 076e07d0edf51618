set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_rl; mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p -- python3 $R/scripts/eval_only.py lstm 500 > $O/p1.log 2>&1 && python3 $R/scripts/pmc_kernel.py $O/p1/p_counter_collection.csv rnn_large_step_bf3 && python3 $R/scripts/pmc_kernel.py $O/p1/p_counter_collection.csv "ce_core_bf3_kernel<13, false, false>" && \
timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/p2 -o p -- python3 $R/scripts/eval_only.py lstm 500 > $O/p2.log 2>&1 && python3 $R/scripts/pmc_kernel.py $O/p2/p_counter_collection.csv rnn_large_step_bf3
rm -f $O/p1/p_counter_collection.csv $O/p2/p_counter_collection.csv
tail -3 $O/p2.log
