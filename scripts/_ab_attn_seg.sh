timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vit_kernels or vit_ti_against or vitc" 2>&1 | tail -3
for i in 1 2 3; do python scripts/probe/attn_probe.py; BCOS_HIP_LIB=$PWD/b-cosification_amd/lib/variants/noseg.so python scripts/probe/attn_probe.py; done
