for v in ${VARIANTS:-default}; do
  if [ $v != default ]; then export BCOS_HIP_LIB=$GRAFT_REPO_ROOT/b-cosification_amd/lib/variants/$v.so; else unset BCOS_HIP_LIB; fi
  echo "== $v"; PSHAPES=${PSH:-3} timeout 200 python scripts/d_bench.py 2>&1 | grep fwd | cut -c1-62
done
