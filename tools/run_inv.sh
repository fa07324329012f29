export ONLY=256:64:64:3,256:64:192:1,256:64:64:1,256:64:32:1,128:128:128:3,64:256:256:3,32:512:512:3,8:1024:1024:3,256:128:64:3,128:128:384:1,16:512:512:3
for d in 0 32 8 11; do echo "== dbg $d"; timeout -k 10 120 python tools/gemm_inventory.py 16 256 $d 2>&1 | grep -v amdgpu.ids | awk '!seen[$3$4$5$6]++' ; done
