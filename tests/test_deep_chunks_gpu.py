"""A frozen MiT encoder runs its deep stages (2-4) as two concurrent chains over slices of the batch (backbones/mit.py::forward_features, round 6;
reference mix_transformer.py:336-365 runs the four stages in sequence): same features as the one-chain forward, eager and inside a captured
hipGraph (the slices become parallel branches), and never when a hook watches a module of those stages or a graph is being built for autograd."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _net(seed=0):
    import segdistill_amd
    from segdistill_amd.builder import BACKBONES, build_from_cfg
    segdistill_amd.register_all()
    torch.manual_seed(seed)
    return build_from_cfg(dict(type='mit_b1'), BACKBONES).to(DEV).eval()


@pytest.mark.parametrize('amp', [True, False])
def test_sliced_deep_stages_match_the_single_chain(amp, monkeypatch):
    from segdistill_amd.backbones import mit
    net = _net()
    x = torch.randn(4, 3, 128, 128, device=DEV)
    monkeypatch.setattr(mit, '_DEEP_CHUNKS_F32', True)
    outs = {}
    for n in (1, 2):
        monkeypatch.setattr(mit, 'DEEP_CHUNKS', n)
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            outs[n] = [f.float().clone() for f in net(x)]
        torch.cuda.synchronize()
    for a, b in zip(outs[1], outs[2]):
        assert a.shape == b.shape
        tol = 2e-2 if amp else 2e-5          # the Linears may take another tile (another summation order) at half the tokens
        assert float((a - b).abs().max()) <= tol * (float(a.abs().max()) + 1e-6)
    assert torch.equal(outs[1][0], outs[2][0])       # stage 1 is not sliced


def test_sliced_forward_inside_a_captured_graph_and_the_guards(monkeypatch):
    from segdistill_amd.backbones import mit
    net = _net(1)
    monkeypatch.setattr(mit, '_DEEP_CHUNKS_F32', True)
    monkeypatch.setattr(mit, 'DEEP_CHUNKS', 2)
    x = torch.randn(4, 3, 128, 128, device=DEV)
    with torch.no_grad():
        ref = [f.clone() for f in net(x)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net(x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = net(x)
        x.copy_(torch.randn(4, 3, 128, 128, device=DEV))
        g.replay()
        torch.cuda.synchronize()
        ref2 = [f.clone() for f in net(x)]
    for a, b in zip(out, ref2):
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-6)
    assert any(float((a - b).abs().max()) > 1e-3 for a, b in zip(ref, ref2))       # the replay really saw the new image
    # guards: a hooked module of the deep stages, an odd batch, autograd
    calls = []
    real = torch.Tensor.chunk
    monkeypatch.setattr(torch.Tensor, 'chunk', lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1])
    h = net.block3[0].register_forward_hook(lambda m, i, o: None)
    with torch.no_grad():
        net(x)
    h.remove()
    with torch.no_grad():
        net(x[:3])
    net(x)                                # grad mode on: the student's path
    assert not calls
    with torch.no_grad():
        net(x)
    assert calls
