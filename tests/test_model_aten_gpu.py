"""GPU: tests/aten_reference.py (PyTorch-ROCm conv stack + HIP decoder, a test instrument) against the reference's golden
vectors.  This pins the boundary wiring (state_dict -> forward) independently of the native conv engine."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["tiny_instance_softmax.npz", "tiny_instance_sum.npz", "tiny_batch_softmax.npz",
                                  "tiny_instance_softmax_k5.npz", "tiny_batch_softmax_k7.npz"])
def test_aten_backend_tiny_golden(golden_dir, name):
    from pixelwiseregression_amd import PixelwiseRegression
    g = np.load(os.path.join(golden_dir, name))
    dev = torch.device("cuda:0")
    m = PixelwiseRegression(int(g["cfg_joints"]), stage=int(g["cfg_stage"]), label_size=int(g["cfg_label_size"]),
                            features=int(g["cfg_features"]), level=int(g["cfg_level"]), kernel_size=int(g["cfg_kernel_size"]),
                            norm_method=str(g["cfg_norm_method"]), heatmap_method=str(g["cfg_heatmap_method"]))
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}, strict=True)
    m = m.to(dev)
    from aten_reference import aten_forward
    batch = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    res = aten_forward(m, batch["img"], batch["label_img"], batch["mask"])
    loss = 0
    for s, (p, D, uvd) in enumerate(res):
        np.testing.assert_allclose(uvd.detach().cpu().numpy(), g["a100_s%d_uvd" % s], atol=5e-4)
        np.testing.assert_allclose(p.detach().cpu().numpy(), g["a100_s%d_p" % s], atol=5e-4)
        loss = loss + torch.mean(torch.sum((uvd - batch["uvd"]) ** 2, dim=2))
    loss.backward()
    assert abs(loss.item() - float(g["a100_loss"])) < 1e-3
    # MIOpen's fp32 conv backward is itself percents away from the CPU path on this ill-conditioned tiny model
    # (see test_engine_gpu.py): direction and size of the flat gradient only
    ge = np.concatenate([(p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(p.shape, np.float32)).ravel()
                         for _, p in m.named_parameters()]).astype(np.float64)
    gr = np.concatenate([g["a100_grad_" + k].ravel() for k, _ in m.named_parameters()]).astype(np.float64)
    cos = float(ge @ gr / (np.linalg.norm(ge) * np.linalg.norm(gr)))
    assert cos > 0.99, cos
