"""csrc/resize.hip (bilinear resize of NCHW maps, forward + deterministic gather backward) against F.interpolate in fp64 on the CPU:
up- and down-scaling, non-integer and anisotropic factors, both align_corners conventions, 1-pixel maps, widths that are not a multiple of
4 (scalar store path), bf16 storage; and the `layers.resize` wrapper's routing."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, C, h, w, H, W
    (2, 5, 16, 16, 64, 64), (1, 3, 6, 6, 16, 16), (2, 4, 1, 1, 8, 8), (1, 7, 2, 3, 64, 64), (2, 3, 33, 20, 128, 100), (1, 2, 40, 56, 128, 128),
    (1, 4, 64, 64, 16, 16), (2, 2, 30, 50, 17, 23), (1, 3, 16, 16, 128, 128), (1, 2, 17, 9, 17, 31), (1, 6, 3, 3, 512, 512),
    # integer factors 2 / 4 / 8 (round 4: the gap-wise forward and the one-kernel LDS backward): several bands per plane with a ragged last
    # one, one-row and one-column maps, widths whose rows are / are not 16-byte multiples (the latter stay on the generic kernels)
    (2, 3, 64, 64, 128, 128), (1, 2, 7, 10, 14, 20), (1, 3, 5, 6, 40, 48), (3, 2, 33, 12, 132, 48), (1, 2, 64, 64, 256, 256), (1, 2, 1, 16, 2, 32),
    (1, 2, 16, 1, 128, 8), (1, 1, 100, 8, 200, 16), (1, 2, 3, 3, 6, 6),
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('align', [False, True])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_bilinear_matches_interpolate(case, align, dtype):
    from segdistill_amd import resize
    B, C, h, w, H, W = case
    g = torch.Generator().manual_seed(h * 7 + w + H)
    x = torch.randn(B, C, h, w, generator=g).to(dtype)
    up = torch.randn(B, C, H, W, generator=g).to(dtype)
    x64 = x.double().requires_grad_(True)
    ref = F.interpolate(x64, size=(H, W), mode='bilinear', align_corners=align)
    ref.backward(up.double())
    dev = torch.device('cuda:0')
    xg = x.to(dev).requires_grad_(True)
    assert resize.supported(xg, (H, W), 'bilinear', align)
    y = resize.bilinear(xg, (H, W), align)
    y.backward(up.to(dev))
    tol = 2e-5 if dtype == torch.float32 else 1e-2        # fp32 source-index arithmetic (as ATen's) against the fp64 reference's: lambda off by ~1e-6
    assert y.shape == ref.shape and y.dtype == dtype
    assert float((y.double().cpu() - ref.detach()).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    gref = x64.grad
    assert float((xg.grad.double().cpu() - gref).abs().max()) <= (5e-5 if dtype == torch.float32 else 2e-2) * max(1.0, float(gref.abs().max()))


def test_backward_is_the_exact_transpose_of_the_forward():
    """<resize(x), u> == <x, resize^T(u)> in fp64-accumulated dot products: the gather backward re-evaluates the forward's own taps."""
    from segdistill_amd import resize
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(4)
    for (h, w, H, W, align) in [(13, 29, 64, 47, False), (64, 64, 128, 128, False), (50, 31, 20, 77, True), (16, 16, 128, 128, True)]:
        x = torch.randn(2, 3, h, w, device=dev, generator=g, requires_grad=True)
        u = torch.randn(2, 3, H, W, device=dev, generator=g)
        y = resize.bilinear(x, (H, W), align)
        y.backward(u)
        lhs, rhs = float((y.double() * u.double()).sum()), float((x.detach().double() * x.grad.double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs)), (h, w, H, W, align, lhs, rhs)


def test_layers_resize_routes_contiguous_cuda_maps_and_keeps_aten_elsewhere():
    from segdistill_amd import layers
    dev = torch.device('cuda:0')
    x = torch.randn(2, 8, 16, 16, device=dev, requires_grad=True)
    y = layers.resize(x, size=(64, 64), mode='bilinear', align_corners=False)
    assert type(y.grad_fn).__name__ == '_BilinearBackward'
    ref = F.interpolate(x, size=(64, 64), mode='bilinear', align_corners=False)
    assert float((y - ref).abs().max()) < 2e-6          # against ATen's own fp32 kernel: the same index arithmetic
    same = layers.resize(x, size=torch.Size((16, 16)), mode='bilinear')                          # same size: a fresh tensor, as F.interpolate
    assert same is not x and same.data_ptr() != x.data_ptr() and torch.equal(same, x)
    assert layers.resize(x, size=(16, 16), mode='bilinear', alias_ok=True) is x                    # read-only callers (the criteria) may alias
    view = torch.randn(2, 16, 16, 16, device=dev)[:, ::2]                                          # strided, not channels-last: copied, then the HIP kernel
    yv = layers.resize(view, size=(48, 40), mode='bilinear', align_corners=False)
    assert float((yv - F.interpolate(view, size=(48, 40), mode='bilinear', align_corners=False)).abs().max()) < 2e-6
    cl = x.detach().contiguous(memory_format=torch.channels_last)
    assert float((layers.resize(cl, size=(32, 32), mode='bilinear', align_corners=False) - F.interpolate(cl, size=(32, 32), mode='bilinear',
                                                                                                          align_corners=False)).abs().max()) == 0.0
    near = layers.resize(x.detach(), size=(32, 32), mode='nearest')
    assert torch.equal(near, F.interpolate(x.detach(), size=(32, 32), mode='nearest'))
    cpu = layers.resize(x.detach().cpu(), size=(32, 32), mode='bilinear', align_corners=False)
    assert not cpu.is_cuda
