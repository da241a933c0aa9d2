#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 1200 python3 -m pytest tests/test_config4_gpu.py tests/test_fresh_device_gpu.py -x -q 2>&1 | tail -3
MAP_MODE=fresh_device RING=32 K=124 timeout 600 python3 tools/time_config4.py 2>&1 | tail -1 | cut -c1-160
MAPS=96 K=200 timeout 600 python3 tools/time_config4.py 2>&1 | tail -1 | cut -c1-160
MEMO=1 K=300 timeout 600 python3 tools/time_config4.py 2>&1 | tail -1 | cut -c1-160
