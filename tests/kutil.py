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
        rl = float(err.norm() / max(float(ref.norm()), 1e-30))
        if rl > rel_l2:
            raise AssertionError("%s: relative L2 error %.4g > %.4g" % (name, rl, rel_l2))
    bound = atol + rtol * ref.abs()
    bad = err > bound
    if float(bad.sum()) > outlier_frac * bad.numel():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError("%s: %d/%d mismatches, max err %.4g (ref max %.4g), first at %s got %.6g ref %.6g" %
                             (name, int(bad.sum()), bad.numel(), float(err.max()), float(ref.abs().max()), idx,
                              float(got[tuple(idx)]), float(ref[tuple(idx)])))
