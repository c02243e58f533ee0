mkdir -p gpurun_out/r06j
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06j/pytest.txt; cat gpurun_out/r06j/pytest.txt
bash scripts/_ab_lib.sh "b-cosification_amd/lib/variants/ns3.so" 3 > gpurun_out/r06j/ab.txt 2>&1; cat gpurun_out/r06j/ab.txt
one() { env $1 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$3', r['value'], r['ms_per_step'])"; }
for i in 1 2; do
one BCOS_NOOP=1 "--arch clip_rn50 --forward-only" "clip-fwd"
one BCOS_HIP_LIB=b-cosification_amd/lib/variants/ns3.so "--arch clip_rn50 --forward-only" "clip-fwd-ns3"
one BCOS_NOOP=1 "--arch vit_ti --batch 512" "vit"
one BCOS_HIP_LIB=b-cosification_amd/lib/variants/ns3.so "--arch vit_ti --batch 512" "vit-ns3"
done
