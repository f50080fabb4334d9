"""Space-to-depth stem: CPU check of the filter / input regrouping used by the GPU path."""
import torch


def test_s2d_filter_regrouping_is_the_same_convolution():
    """CPU check of the index algebra: conv7x7/2/p3(x, w) == conv4x4/1/p0(s2d(x), s2d_filter(w))."""
    from distributed_deep_learning_amd.ops.kernels import s2d_filter
    torch.manual_seed(0)
    x = torch.randn(2, 3, 20, 12, dtype=torch.float64)
    w = torch.randn(5, 3, 7, 7, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(x, w, stride=2, padding=3)
    N, _, H, W = x.shape
    xp = torch.nn.functional.pad(x, (3, 3 + 2, 3, 3 + 2))                 # index i of xp <-> image row i - 3
    s2d = torch.zeros(N, 16, H // 2 + 3, W // 2 + 3, dtype=x.dtype)
    for p in range(2):
        for q in range(2):
            for c in range(3):
                # out[h2][w2] = x[2*(h2-1)+p-1][2*(w2-1)+q-1] = xp[2*h2 + p][2*w2 + q]
                s2d[:, (p * 2 + q) * 3 + c] = xp[:, c, p::2, q::2][:, :H // 2 + 3, :W // 2 + 3]
    out = torch.nn.functional.conv2d(s2d, s2d_filter(w))
    assert out.shape == ref.shape and torch.allclose(out, ref, atol=1e-10)
