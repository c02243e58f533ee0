python - <<'PY'
import sys, os
sys.path.insert(0, "b-cosification_amd")
import torch
from bcos_hip import lib, ops
lib.load()
torch.manual_seed(0)
dev = torch.device("cuda", 0)
for (M, K, N) in [(50176, 1024, 256), (12544, 2048, 512), (200704, 512, 128), (50176, 256, 1024), (3000, 512, 384)]:
    a = torch.randn(M, K, device=dev); ops.ensure_absmax(a.view(1, 1, M, K))
    w = ops.mark_static(torch.randn(N, K, device=dev) * 0.05)
    outs = []
    for tile in (0, 3):
        lib.set_option("h2_tile", tile)
        a4 = a.view(1, 1, M, K); ops.ensure_absmax(a4)
        o = ops.matmul_nt(a4.view(M, K) if False else a, w) if False else None
        out = torch.empty(1, 1, M, N, device=dev)
        g = ops.fwd_geom(1, 1, M, K, N, 1, 1, 1, 1, 0, 0)
        ops.tapconv(a4, w.view(N, 1, 1, K), g, out=out, bcos_mode=lib.BCOS_CONV_EPS, b=2.0, relu=True, scale_out=torch.empty_like(out))
        outs.append(out.clone())
    lib.set_option("h2_tile", 0)
    print((M, K, N), "bit-identical:", torch.equal(outs[0], outs[1]), float(outs[0].abs().max()))
PY
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-ref 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); rf=r['roofline']; print('$1', r['value'], r['step_times']['sub_batch_stream_steps']['median'], rf['kernel_ms_per_step'], rf['by_bound']['mfma']['ms_per_step'], rf['by_bound']['hbm']['ms_per_step'])"; }
for i in 1 2 3; do
  run BCOS_NOOP=1
  run BCOS_OPT_H2_TILE=3
done
