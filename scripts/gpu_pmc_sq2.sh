#!/bin/bash
./scripts/gpu_pmc_sq.sh rne 2>&1 | grep "k_rne" | grep "SQ_WAVES\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU\|SQ_ACTIVE_INST_VALU\|SQ_WAIT_INST_ANY\|SQ_ACTIVE_INST_ANY\|GRBM_GUI"
EXTRA_ARGS="--n-ik 1000000" ./scripts/gpu_pmc_sq.sh ik 2>&1 | grep "k_ik" | grep "SQ_WAVES\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU\|SQ_ACTIVE_INST_VALU\|SQ_WAIT_INST_ANY\|SQ_ACTIVE_INST_ANY\|GRBM_GUI"
./scripts/gpu_pmc_sq.sh dyn 2>&1 | grep "k_dyn" | grep "SQ_WAVES\|SQ_BUSY_CYCLES\|SQ_INSTS_VALU\|SQ_ACTIVE_INST_VALU\|GRBM_GUI"
