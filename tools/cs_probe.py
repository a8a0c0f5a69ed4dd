import sys, torch
sys.path.insert(0, "/root/repo")
from anyedit_amd import ops
DEV, BF = "cuda", torch.bfloat16
B = 12
def timeit(fn, iters=40, warm=6):
    for i in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (H, Cin, Cout) in ((64, 320, 320), (64, 960, 320), (32, 640, 640), (32, 1920, 640)):
    M = B * H * H
    ko = ops.conv_k_order(M, Cin, Cout)
    w = ops.pack_conv3x3(torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.02, k_order=ko)
    x = torch.randn(M, Cin, device=DEV).to(BF); bias = torch.randn(Cout, device=DEV); emb = torch.randn(B, Cout, device=DEV)
    res = torch.randn(M, Cout, device=DEV).to(BF)
    cs = ops.colstats_buffer(M, Cout, DEV)
    t0 = timeit(lambda: ops.conv3x3(x, w, bias, B, H, H, k_order=ko))
    t1 = timeit(lambda: ops.conv3x3(x, w, bias, B, H, H, k_order=ko, colstats=cs))
    t2 = timeit(lambda: ops.conv3x3(x, w, bias, B, H, H, k_order=ko, addvec=emb, colstats=cs))
    t3 = timeit(lambda: ops.conv3x3(x, w, bias, B, H, H, k_order=ko, residual=res, colstats=cs))
    print(f"conv {Cin}->{Cout} @{H}: plain {t0:.1f} us, +colstats {t1:.1f}, +addvec+colstats {t2:.1f}, +residual+colstats {t3:.1f}", flush=True)
