for v in ${VARIANTS:-default}; do
  if [ $v != default ]; then export BCOS_HIP_LIB=$GRAFT_REPO_ROOT/b-cosification_amd/lib/variants/$v.so; else unset BCOS_HIP_LIB; fi
  echo "== $v"; timeout 200 python scripts/stem_dgrad_bench.py 2>&1 | grep -E "patch"
done
