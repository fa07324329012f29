export ONLY=128:256:64:3,64:512:128:3,128:128:128:3,64:256:256:3,32:512:512:3,256:128:64:3,256:64:64:3
for d in 0 1 2 8; do echo "== dbg $d"; timeout -k 10 120 python tools/gemm_inventory.py 16 256 $d 2>&1 | grep -v amdgpu.ids | awk '!seen[$3$4$5$6]++' ; done
