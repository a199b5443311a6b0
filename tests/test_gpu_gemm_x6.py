"""bf16x6 GEMM (buctd_x6_image + buctd_x6_gemm) against fp64: plain, transposed, grouped operands and ragged sizes,
the fc_o = nn.Linear(T, T) products of the CoAM channel attention (self_attention.py:150-159)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def rel_err(got, ref):
    return (got.double().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)


@pytest.mark.parametrize("M,N,K", [(128, 192, 128), (200, 300, 100), (16, 16, 32), (257, 193, 321), (512, 384, 1024)])
def test_plain_and_transposed_operands(dev, M, N, K):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ B.double()
    Ad, Bd = A.to(dev), B.to(dev)
    a_img = ops.x6_image(Ad, M, K, 0, vs=K, ks=1)                    # A row-major
    b_img = ops.x6_image(Bd, N, K, 1, vs=1, ks=N)                    # B row-major [K][N]: v = n, k strides N
    out = torch.full((M, N), float("nan"), device=dev)
    ops.x6_gemm(a_img, b_img, out, M, N, K, ldc=N, bias=bias.to(dev), bias_axis=0, alpha=0.5)
    assert rel_err(out, 0.5 * ref + bias.double()) <= 2e-6
    # the same product from transposed storage: A^T stored [K][M], B^T stored [N][K]
    At, Bt = A.t().contiguous().to(dev), B.t().contiguous().to(dev)
    a_img = ops.x6_image(At, M, K, 0, vs=1, ks=M)
    b_img = ops.x6_image(Bt, N, K, 1, vs=K, ks=1)
    out2 = torch.full((M, N), float("nan"), device=dev)
    rb = torch.randn(M, generator=g)
    ops.x6_gemm(a_img, b_img, out2, M, N, K, ldc=N, bias=rb.to(dev), bias_axis=1)
    assert rel_err(out2, ref + rb.double()[:, None]) <= 2e-6


def test_exact_on_adversarial_operands(dev):
    """all 24 mantissa bits set, magnitudes over 2^+-12 inside one reduction: the six-term product keeps fp32 class"""
    from buctd_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 128, 192, 256
    mant = (torch.randint(0, 2 ** 23, (M, K), generator=g) | 1).float() / 2 ** 23 + 1.0
    A = mant * torch.exp2(torch.randint(-12, 13, (M, K), generator=g).float()) * (torch.randint(0, 2, (M, K), generator=g) * 2 - 1)
    mant = (torch.randint(0, 2 ** 23, (K, N), generator=g) | 1).float() / 2 ** 23 + 1.0
    B = mant * torch.exp2(torch.randint(-12, 13, (K, N), generator=g).float())
    ref = A.double() @ B.double()
    scale = (A.double().abs() @ B.double().abs())
    out = torch.empty(M, N, device=dev)
    ops.x6_gemm(ops.x6_image(A.to(dev), M, K, 0, vs=K, ks=1), ops.x6_image(B.to(dev), N, K, 1, vs=1, ks=N), out, M, N, K, ldc=N)
    assert ((out.double().cpu() - ref).abs() / scale).max().item() <= 1e-6


def test_fc_o_grouped_layouts(dev):
    """the three fc_o products on token-major activations on [B][T][C]: forward (n-grouped B operand and output),
    data gradient (transposed weight), weight gradient (k-grouped operands)."""
    from buctd_amd import ops
    g = torch.Generator().manual_seed(11)
    Bn, T, Cn = 3, 160, 48
    W = torch.randn(T, T, generator=g) * T ** -0.5
    b = torch.randn(T, generator=g)
    on = torch.randn(Bn, T, Cn, generator=g)
    dout = torch.randn(Bn, T, Cn, generator=g)
    Wd, ond, doutd = W.to(dev), on.to(dev), dout.to(dev)
    # out[b][t'][c] = sum_t W[t'][t] on[b][t][c] + bias[t']
    ref = torch.einsum("pt,btc->bpc", W.double(), on.double()) + b.double()[None, :, None]
    w_img = ops.x6_image(Wd, T, T, 0, vs=T, ks=1)
    on_img = ops.x6_image(ond, Bn * Cn, T, 1, vg=Cn, vgs=T * Cn, vs=1, ks=Cn)
    out = torch.empty_like(ond)
    ops.x6_gemm(w_img, on_img, out, T, Bn * Cn, T, ldc=Cn, Nc=Cn, gsc=T * Cn, bias=b.to(dev), bias_axis=1)
    assert rel_err(out, ref) <= 2e-6
    # d_on[b][t][c] = sum_t' W[t'][t] dout[b][t'][c]
    ref = torch.einsum("pt,bpc->btc", W.double(), dout.double())
    wt_img = ops.x6_image(Wd, T, T, 0, vs=1, ks=T)
    do_img = ops.x6_image(doutd, Bn * Cn, T, 1, vg=Cn, vgs=T * Cn, vs=1, ks=Cn)
    don = torch.empty_like(ond)
    ops.x6_gemm(wt_img, do_img, don, T, Bn * Cn, T, ldc=Cn, Nc=Cn, gsc=T * Cn)
    assert rel_err(don, ref) <= 2e-6
    # dW[t'][t] = sum_{b,c} dout[b][t'][c] on[b][t][c]
    ref = torch.einsum("bpc,btc->pt", dout.double(), on.double())
    a_img = ops.x6_image(doutd, T, Bn * Cn, 0, vs=Cn, ks=1, kg=Cn, kgs=T * Cn)
    b_img = ops.x6_image(ond, T, Bn * Cn, 1, vs=Cn, ks=1, kg=Cn, kgs=T * Cn)
    dW = torch.empty(T, T, device=dev)
    ops.x6_gemm(a_img, b_img, dW, T, T, Bn * Cn, ldc=T)
    assert rel_err(dW, ref) <= 2e-6
