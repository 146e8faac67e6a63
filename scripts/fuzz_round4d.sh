#!/bin/bash
# The fuzzer of scripts/fuzz_round4c.sh that stopped at a degenerate input (case 86 of seed 474747: ACH of a cube, refused at
# upload; scripts/fuzz_gpu.py now counts it as outside the reference's domain), once more.  Run through gpurun.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
export SURTR_EVENTS_IN_FLIGHT=6
python scripts/fuzz_gpu.py 2500 474747 > gpurun_out/fuzz/r4g_fuzz_474747.log 2>&1; tail -1 gpurun_out/fuzz/r4g_fuzz_474747.log
