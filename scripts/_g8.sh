#!/bin/bash
cd $GRAFT_REPO_ROOT
env | grep -E "HIP|ROCR|HSA|CUDA|GPU" 
for m in create event torch_first; do python scripts/_diag_torch.py $m 2>&1 | tail -12; done
