#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export SSG_LIB_PATH=$ROOT/ship_sim_gym_amd/libshipsim_stamps.so
python3 tools/stamps_single.py 2>&1 | tail -6
SSG_SHIPS=4 SSG_NB=10 python3 tools/stamps_single.py 2>&1 | tail -6
