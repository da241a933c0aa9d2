#!/bin/bash
# Development aid: build a VARIANT of libshipsim.so (diagnostic stamps, ablation switches, ...) next to the product
# library without touching it:   tools/build_variant.sh stamps -DSSG_STAMPS   ->  ship_sim_gym_amd/libshipsim_stamps.so
# Use it with SSG_LIB_PATH=ship_sim_gym_amd/libshipsim_stamps.so (read by ship_sim_gym_amd/_native.py).
set -e
TAG=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/ssg_variant_$TAG
mkdir -p $B
cp $REPO/ship_sim_gym_amd/csrc/*.hip $REPO/ship_sim_gym_amd/csrc/*.cpp $REPO/ship_sim_gym_amd/csrc/*.h $REPO/ship_sim_gym_amd/csrc/Makefile $B/
make -C $B -s -j8 INC="-I$REPO/include -I$B" HDRS= OUT=$REPO/ship_sim_gym_amd/libshipsim_$TAG.so GROUPS="${SSG_GROUPS:-1 2}" EXTRA="$*"
ls -la $REPO/ship_sim_gym_amd/libshipsim_$TAG.so
