# experiment switches live in the -DDDPMIR_EXPERIMENTS build only (python -m ddpm_image_restoration_amd.build --experiments)
export DDPMIR_LIB_PATH=${DDPMIR_LIB_PATH:-$(dirname $0)/../ddpm_image_restoration_amd/libddpmir_exp.so}
# runs the model-level GPU parity tests under every experiment switch (the alternative paths must stay correct)
for v in DDPMIR_NO_QKV_FUSE DDPMIR_SPLIT_GATES DDPMIR_SPLIT_SHORTCUT DDPMIR_NO_GN_FUSE DDPMIR_GEMM_NO_LONGK DDPMIR_NO_OVERLAP; do
  echo "== $v=1"; env $v=1 timeout -k 10 300 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -1
done
