cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
R=128 K=635 python3 tools/time_fresh_device.py
SSG_LIB_PATH=$PWD/ship_sim_gym_amd/libshipsim_genw2.so R=128 K=635 python3 tools/time_fresh_device.py
done
SSG_LIB_PATH=$PWD/ship_sim_gym_amd/libshipsim_genw2.so R=128 K=635 bash tools/fresh_kernel_times.sh
