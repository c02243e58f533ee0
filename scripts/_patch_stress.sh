for i in 1 2 3 4 5 6 7 8; do timeout 600 python scripts/probe/patch_stress.py $i 2>&1 | grep -v amdgpu.ids | head -12 & done; wait
echo "--- control: BCOS_OPT_PATCH=0"
for i in 1 2 3 4 5 6 7 8; do BCOS_OPT_PATCH=0 timeout 600 python scripts/probe/patch_stress.py $i 2>&1 | grep -v amdgpu.ids | head -4 & done; wait
