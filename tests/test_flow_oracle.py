"""CPU checks of oracle/flow_oracle.py (the chained-flow checker of tests/test_gpu_chained_flow.py): its forward in 'fp' mode is the
model's own forward (+ the PixelShuffle wrapper's LeakyReLU on the decoder output), its caches are the hooked tensors,
and a short chained run leaves every unit trained with hard rounding."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


@pytest.mark.parametrize("arch", ["cheng", "minnen"])
def test_flow_oracle_fp_forward_and_caches(arch):
    from oracle import lic_oracle as L
    from oracle.flow_oracle import FlowOracle
    torch.manual_seed(3)
    model = (L.Cheng2020Anchor(N=8) if arch == "cheng" else L.MeanScaleHyperprior(N=8, M=12)).eval()
    if arch == "cheng":
        model.context_prediction.weight.data *= model.context_prediction.mask
    x = torch.rand(2, 3, 64, 64)
    flow = FlowOracle(model)
    assert len(flow.units) == (29 if arch == "cheng" else 20)
    flow._set_modes("fp")
    with torch.no_grad():
        want = model(x)
        got = flow.forward(x)
    xh = F.leaky_relu(want["x_hat"], 0.01) if arch == "cheng" else want["x_hat"]
    torch.testing.assert_close(got["x_hat"], xh, rtol=1e-5, atol=1e-6)
    for k in ("y", "z"):
        torch.testing.assert_close(got["likelihoods"][k], want["likelihoods"][k], rtol=1e-4, atol=1e-7)
    name = "g_a.1"
    io = {}
    h = dict(model.named_modules())[name].register_forward_hook(lambda m, i, o: io.update(i=i[0].clone(), o=o.clone()))
    with torch.no_grad():
        model(x)
    h.remove()
    xq, xf, tg = flow.caches(name, x, batch=1)
    torch.testing.assert_close(xf, io["i"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(tg, io["o"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(xq, xf)          # nothing trained yet: the "quantised prefix" is still full precision


def test_flow_oracle_short_chain():
    from oracle import lic_oracle as L
    from oracle.flow_oracle import FlowOracle
    torch.manual_seed(4)
    model = L.MeanScaleHyperprior(N=4, M=6).eval()
    cali = torch.rand(4, 3, 64, 64)
    flow = FlowOracle(model)
    idx = {u.name: np.stack([np.random.RandomState(i).permutation(4)[:2] for i in range(3)]) for u in flow.units}
    seen = []
    logs = flow.recon_model(cali, idx, 1005, iters=3, batch_size=2, on_unit=lambda u: seen.append(u.name))
    assert seen == [u.name for u in flow.units] and all(u.trained for u in flow.units)
    assert all(np.isfinite(l.total).all() for l in logs.values())
    # the second unit was calibrated on the quantised output of the first: its x_q differs from x_fp now
    xq, xf, _ = flow.caches(flow.units[1].name, cali)
    assert float((xq - xf).abs().max()) > 0
    psnr, bpp = flow.evaluate([torch.rand(1, 3, 72, 100)], p=64, act_quant=True)
    assert np.isfinite(psnr) and bpp > 0
