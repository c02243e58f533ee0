import torch, torch.nn.functional as F
def rel(a, b): return ((a.double().cpu()-b.double().cpu()).norm()/b.double().cpu().norm()).item()
torch.manual_seed(0)
x = torch.randn(2, 16, 10, 10)
for name, fn in (("avg3", lambda t: F.avg_pool2d(t, 3, 2, 1)), ("adaptive", lambda t: F.adaptive_avg_pool2d(t, 1)), ("relu", torch.relu)):
    xr = x.clone().requires_grad_(True); yr = fn(xr); gy = torch.randn(yr.shape); (gr,) = torch.autograd.grad(yr, xr, gy)
    for xin_cl in (False, True):
        for g_cl in (False, True):
            xg = x.clone().cuda()
            if xin_cl: xg = xg.contiguous(memory_format=torch.channels_last)
            xg.requires_grad_(True)
            y = fn(xg)
            g = gy.cuda()
            if g_cl: g = g.contiguous(memory_format=torch.channels_last)
            (gx,) = torch.autograd.grad(y, xg, g)
            print(name, "x_cl", xin_cl, "g_cl", g_cl, "y", rel(y, yr), "gx", rel(gx, gr), "y strides", y.stride())
