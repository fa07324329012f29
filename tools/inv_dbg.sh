export DDPMIR_LIB_PATH=$GRAFT_REPO_ROOT/ddpm_image_restoration_amd/libddpmir_exp.so
export ONLY=256:64:64:3,256:128:64:3,128:64:64:3,128:256:64:3,128:128:128:3,64:256:256:3,64:512:128:3,32:512:512:3,32:1024:256:3
for d in 0 1 2 3; do echo "== dbg $d (1 = no A loads, 2 = no B loads, 3 = neither)"; timeout -k 10 200 python tools/gemm_inventory.py 16 256 $d 2>&1 | grep -v amdgpu.ids | awk '!seen[$3$4$5$6]++' ; done
