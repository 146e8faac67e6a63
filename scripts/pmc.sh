#!/bin/bash
# Collects PMC counters for one bench run per pass (rocprofv3 --pmc only; no other trace domains).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1; echo "$name exit=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
find $OUT -name "*counter_collection.csv" | head
