"""Helpers shared by the GPU kernel parity tests: NHWC-padded buffers <-> NCHW tensors, tolerances."""
import torch


def pad8(c):
    return (c + 7) // 8 * 8


def to_act(x_nchw, dtype, ld=None, device="cuda"):
    """NCHW (any float dtype, CPU) -> [N*H*W, ld] buffer on the GPU, padding channels zero."""
    N, C, H, W = x_nchw.shape
    ld = ld or pad8(C)
    buf = torch.zeros(N * H * W, ld, dtype=dtype, device=device)
    buf[:, :C] = x_nchw.permute(0, 2, 3, 1).reshape(-1, C).to(dtype).to(device)
    return buf


def from_act(buf, N, H, W, C):
    """[M, ld] GPU buffer -> NCHW float64 CPU tensor."""
    return buf[:, :C].double().cpu().reshape(N, H, W, C).permute(0, 3, 1, 2).contiguous()


def rounded(x_nchw, dtype):
    """Value of x after storage in `dtype` (so references see exactly what the kernel reads)."""
    return x_nchw.to(dtype).double()


def cvec(v, device="cuda"):
    """per-channel fp32 vector padded to a multiple of 8"""
    C = v.numel()
    out = torch.zeros(pad8(C), dtype=torch.float32, device=device)
    out[:C] = v.float().to(device)
    return out


def tol(dtype):
    # fp32 kernels: accumulation-order noise only.  bf16 storage: one rounding of the output (2^-8 relative).
    return dict(rtol=2e-4, atol=2e-5) if dtype == torch.float32 else dict(rtol=1.2e-2, atol=1e-2)


def assert_close(name, got, ref, rtol, atol, outlier_frac=0.0, rel_l2=None):
    """Element-wise |got-ref| <= atol + rtol*|ref|, except for at most `outlier_frac` of the elements (bf16 activations
    flip a ReLU mask for pre-activations within rounding distance of zero; such elements are individually wrong by a full
    gradient contribution but rare).  rel_l2 additionally bounds ||got-ref|| / ||ref||."""
    got = got.double().cpu()
    ref = ref.double().cpu()
    err = (got - ref).abs()
    if rel_l2 is not None:
        rl = float(err.norm() / max(float(ref.norm()), float(atol) * ref.numel() ** 0.5, 1e-30))   # atol floors the scale
        if rl > rel_l2:
            raise AssertionError("%s: relative L2 error %.4g > %.4g" % (name, rl, rel_l2))
    bound = atol + rtol * ref.abs()
    bad = err > bound
    if float(bad.sum()) > outlier_frac * bad.numel():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError("%s: %d/%d mismatches, max err %.4g (ref max %.4g), first at %s got %.6g ref %.6g" %
                             (name, int(bad.sum()), bad.numel(), float(err.max()), float(ref.abs().max()), idx,
                              float(got[tuple(idx)]), float(ref[tuple(idx)])))


# ---- deterministic, RNG-free fills shared with tools/make_golden.py (inputs of the golden fixtures are regenerated, not stored)
def counter_fill(t, seed):
    n = t.numel()
    idx = torch.arange(n, dtype=torch.float64)
    v = torch.sin(idx * 12.9898 + seed * 78.233) * 43758.5453
    v = v - torch.floor(v)   # [0, 1)
    return (v - 0.5).reshape(t.shape)


def randomize_counter(module, seed):
    with torch.no_grad():
        for i, (n, p) in enumerate(module.named_parameters()):
            f = counter_fill(p, seed + i)
            if p.dim() == 1 and "bias" not in n:
                p.copy_(f + 1.0)
            elif p.dim() == 1:
                p.copy_(f * 0.4)
            else:
                p.copy_(f * 2.0 / p[0].numel() ** 0.5)
        for i, (n, b) in enumerate(module.named_buffers()):
            if "running_mean" in n:
                b.copy_(counter_fill(b, seed + 1000 + i) * 0.2)
            elif "running_var" in n:
                b.copy_(counter_fill(b, seed + 2000 + i) + 1.0)


def check_digest(name, t, d, rtol=1e-6, atol=0.0):
    """Compares a tensor with the fingerprint stored in a golden fixture (shape, sum, sum of squares, head, tail).  `atol` is an
    absolute per-element floor for tensors that are rounding noise around zero (a BN bias in front of another BN)."""
    f = t.detach().double().cpu().flatten()
    assert tuple(t.shape) == tuple(d["shape"]), (name, tuple(t.shape), d["shape"])
    scale = max(1.0, abs(d["sum"]), d["sumsq"] ** 0.5)
    assert abs(float(f.sum()) - d["sum"]) <= rtol * scale * max(1.0, f.numel() ** 0.5) + atol * f.numel(), (name, float(f.sum()), d["sum"])
    assert abs(float((f * f).sum()) - d["sumsq"]) <= rtol * max(1.0, d["sumsq"]) * 10 + atol * f.numel(), (name, float((f * f).sum()), d["sumsq"])
    s = max(1e-6, float(d["head"].abs().max()), float(d["tail"].abs().max()))
    assert float((f[:4] - d["head"]).abs().max()) <= rtol * 100 * s + 1e-12 + atol, (name, f[:4], d["head"])
    assert float((f[-4:] - d["tail"]).abs().max()) <= rtol * 100 * s + 1e-12 + atol, (name, f[-4:], d["tail"])


def bf16_storage():
    """The oracle's storage model of the bf16 HIP path on THIS library: Bf16Storage, with the depthwise convolutions restated as
    matrix-core ones (fp16 / bf16 MFMA operands, csrc/dwconv_mm.hip) exactly where the library says it runs those kernels."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import atomnas_oracle as orc
    from atomnas_amd import _lib
    lib = _lib.load()

    def pred(k, stride, N, C, H, W, slab):
        if not slab or os.environ.get("ATOMNAS_PLAIN_HIDDEN", "0") != "0":
            return False, False
        return (bool(lib.atomnas_dwconv_mm_supported(N, H, W, C, k, stride, 1, 0)), bool(lib.atomnas_dwconv_mm_supported(N, H, W, C, k, stride, 1, 1)))
    return orc.bf16_storage_mm(pred)
