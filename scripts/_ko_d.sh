for v in default dko1 dko2 dko3; do
  if [ $v != default ]; then export BCOS_HIP_LIB=$GRAFT_REPO_ROOT/b-cosification_amd/lib/variants/$v.so; fi
  echo "== $v"; QUICK=1 timeout 200 python scripts/d_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-60
done
