#!/bin/bash
# round-5 GPU call helper: runs the script body given as $1 (a file under tools/r05/) with the environment every call needs
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
bash "$@"
