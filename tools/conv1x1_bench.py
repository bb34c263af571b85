import time, torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
B = 8
for (h, w, cin, cout) in [(120, 160, 64, 64), (120, 160, 64, 256), (120, 160, 256, 64), (60, 80, 512, 128), (60, 80, 128, 512),
                          (30, 40, 1024, 256), (30, 40, 256, 1024), (15, 20, 2048, 512), (15, 20, 512, 2048)]:
    x = torch.randn(B, cin, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device="cuda")
    w2 = wt.view(cout, cin).t().contiguous()
    def conv(): return F.conv2d(x, wt, bias)
    def mm():
        y = torch.addmm(bias, x.permute(0, 2, 3, 1).reshape(-1, cin), w2)
        return y.view(B, h, w, cout).permute(0, 3, 1, 2)
    a, b_ = conv(), mm()
    err = (a - b_).abs().max().item()
    fl = 2 * B * h * w * cin * cout / 1e9
    tc, tm = t(conv), t(mm)
    print(f"{h}x{w} {cin}->{cout}: conv {tc:.3f} ms ({fl/tc:.1f} TF)  addmm {tm:.3f} ms ({fl/tm:.1f} TF)  err {err:.1e}")
# 3x3 for reference
for (h, w, c) in [(120, 160, 64), (60, 80, 128), (30, 40, 256), (15, 20, 512)]:
    x = torch.randn(B, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = torch.randn(c, c, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    tc = t(lambda: F.conv2d(x, wt, None, 1, 1))
    fl = 2 * B * h * w * c * c * 9 / 1e9
    print(f"3x3 {h}x{w} {c}: {tc:.3f} ms ({fl/tc:.1f} TF)")
