#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python scripts/stamps_wave.py build_tmp/libsurtr_hip_stamp.so 2>&1 | tail -28
