"""Torch-only reproduction attempt of the 8-rank calibration divergence: NO bcos_hip import, no kernel of
this repo in the process -- 8 processes time-slicing one GPU each compute x.var(dim=(0, 2, 3), unbiased=False) of the SAME
deterministic tensors (the shapes of ResNet-50's layer3 BN inputs at 8 images) many times and compare every result bit for
bit with the first one.  A mismatch here shows the wrong variance comes from torch's own multi-block reduction under GPU
time-slicing, not from a stray write of this repo's kernels.
    python scripts/probe/var_stress_torch_only.py [n_procs] [iterations]"""
import sys
import torch
import torch.multiprocessing as mp


def worker(rank, iters, q):
    assert "bcos_hip" not in sys.modules
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(1234)
    shapes = [(8, 1024, 14, 14), (8, 256, 14, 14), (8, 512, 28, 28), (8, 2048, 7, 7)]
    xs = [torch.randn(*s, generator=g).to(dev).contiguous(memory_format=torch.channels_last) for s in shapes]
    ref = [x.var(dim=(0, 2, 3), unbiased=False).clone() for x in xs]
    torch.cuda.synchronize()
    bad = []
    for it in range(iters):
        for i, x in enumerate(xs):
            junk = torch.empty(1 << 22, device=dev).normal_()          # allocator churn + other kernels in between, like a calibration pass
            v = x.var(dim=(0, 2, 3), unbiased=False)
            if not torch.equal(v, ref[i]):
                diff = (v != ref[i]).nonzero().flatten().tolist()
                bad.append((it, i, len(diff), diff[:8], float((v - ref[i]).abs().max())))
            del junk
    torch.cuda.synchronize()
    q.put((rank, bad[:5], len(bad)))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, iters, q)) for r in range(n)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=1200) for _ in ps)
    for p in ps:
        p.join()
    total = sum(r[2] for r in res)
    print(f"torch-only var stress: {n} processes x {iters} iterations x 4 tensors: {total} mismatching reductions")
    for r in res:
        if r[2]:
            print("  rank", r[0], "mismatches", r[2], "first:", r[1])
