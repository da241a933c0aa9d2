#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for i in 1 2; do MEMO=1 timeout 300 python3 tools/time_config4.py 2>&1 | tail -1 | cut -c1-330; done
MEMO=0 timeout 300 python3 tools/time_config4.py 2>&1 | tail -1 | cut -c1-100
