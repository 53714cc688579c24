"""Full-size check of the depthwise entry points against torch's own convolution on the GPU (development aid, not a test):
    python tools/dwcheck.py [N]        cases: DWCHECK_CASES="7,1152,7,1;56,144,3,1" (H,C,k,s)
Forward y and statistics, backward h / dw / statistics, bf16 slab-major tensors, the bench's batch size by default."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import ops
from atomnas_amd.ops import Slab

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CASES = [(56, 144, 3, 1), (56, 144, 5, 1), (56, 144, 7, 1), (28, 240, 3, 1), (28, 240, 5, 1), (28, 240, 7, 1), (14, 480, 3, 1), (14, 480, 7, 1),
         (14, 576, 5, 1), (7, 1152, 3, 1), (7, 1152, 5, 1), (7, 1152, 7, 1)]
if os.environ.get("DWCHECK_CASES"):
    CASES = [tuple(int(v) for v in c.split(",")) for c in os.environ["DWCHECK_CASES"].split(";")]
torch.manual_seed(0)


def nchw(slab, n, h, w, c):
    return slab.to_plain()[:, :c].float().reshape(n, h, w, c).permute(0, 3, 1, 2)


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


for (H, C, k, s) in CASES:
    P = (k - 1) // 2
    Ho = (H - 1) // s + 1
    x2 = torch.randn(N * H * H, C, device="cuda").bfloat16()
    g2 = (torch.randn(N * Ho * Ho, C, device="cuda") * 1e-3).bfloat16()
    w = torch.randn(C, 1, k, k, device="cuda") * 0.3
    taps = w.reshape(C, k * k).t().contiguous()
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.3
    c1, c2, c3 = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 1e-4
    rows = ops.stat_rows_for(C)
    xs, gs = Slab.from_plain(x2), Slab.from_plain(g2)
    ys = Slab(N * Ho * Ho, C, torch.bfloat16, "cuda", zero=True)
    st = torch.full((rows, 2, C), float("nan"), device="cuda")
    ops.dwconv_fwd(xs, sc, sh, True, taps, ys, st, C, N, H, H, C, k, s, stat_rows=rows)
    torch.cuda.synchronize()
    x4 = x2.float().reshape(N, H, H, C).permute(0, 3, 1, 2).requires_grad_(True)
    pre = x4 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    xa = torch.relu(pre)
    wr = w.clone().requires_grad_(True)
    yref = F.conv2d(xa, wr, None, s, P, 1, C)
    y = nchw(ys, N, Ho, Ho, C)
    bad = ~torch.isfinite(y)
    line = "H%-3d C%-4d k%d s%d: fwd rel %.2e nonfinite %d" % (H, C, k, s, rel(torch.nan_to_num(y), yref.detach()), int(bad.sum()))
    if bad.any():
        line += " channels %s" % torch.nonzero(bad.any(0).any(1).any(1)).flatten()[:8].tolist()
    ssum = st.sum(0)
    line += " stat rel %.1e/%.1e" % (rel(ssum[0], y.sum((0, 2, 3))), rel(ssum[1], (y * y).sum((0, 2, 3))))
    # backward: dYraw = c1 g + c2 y + c3 with the kernel's own y
    dy = c1.view(1, -1, 1, 1) * g2.float().reshape(N, Ho, Ho, C).permute(0, 3, 1, 2) + c2.view(1, -1, 1, 1) * y + c3.view(1, -1, 1, 1)
    (yref * dy).sum().backward()
    href = x4.grad   # = dL/dx through relu: conv^T(dy) * (pre > 0) * sc ... x4.grad includes the factor sc; the kernel's h is wrt xa
    hs = Slab(N * H * H, C, torch.bfloat16, "cuda", zero=True)
    dw = torch.zeros(C * k * k, device="cuda")
    st2 = torch.full((rows, 2, C), float("nan"), device="cuda")
    ops.dwconv_bwd(gs, ys, c1, c2, c3, xs, sc, sh, True, taps, hs, dw, st2, C, N, H, H, C, k, s, stat_rows=rows)
    torch.cuda.synchronize()
    h = nchw(hs, N, H, H, C)
    href_h = href / sc.view(1, -1, 1, 1)
    badh = ~torch.isfinite(h)
    line += " | bwd h rel %.2e nonfinite %d dw rel %.2e" % (rel(torch.nan_to_num(h), href_h), int(badh.sum()), rel(torch.nan_to_num(dw.view(C, k * k)), wr.grad.view(C, k * k)))
    s2 = st2.sum(0)
    line += " stat rel %.1e/%.1e" % (rel(torch.nan_to_num(s2[0]), h.sum((0, 2, 3))), rel(torch.nan_to_num(s2[1]), (h * x4.detach()).sum((0, 2, 3))))
    if badh.any():
        line += " channels %s" % torch.nonzero(badh.any(0).any(1).any(1)).flatten()[:8].tolist()
    print(line, flush=True)
    del x4, pre, xa, yref, dy, href
