"""Mid-result cache / checkpoint formats: files a stock install of the reference would read."""
import numpy as np
import torch

from gaussctrl_amd import midcache


def test_view_roundtrip_matches_reference_reader(tmp_path):
    g = torch.Generator().manual_seed(0)
    H = W = 64
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    rgb = torch.stack([xx, yy, 0.5 * (xx + yy)], -1); depth = torch.rand(H, W, generator=g) * 3 + 0.5
    z0 = torch.randn(1, 4, H // 8, W // 8, generator=g); mask = torch.rand(H, W, generator=g) > 0.5
    midcache.save_view(tmp_path, 4, unedited_image=rgb, depth=depth, z_0=z0, mask=mask)
    # exactly what gc_dataparser_ns.py:408-420 looks for (1-based frame numbers)
    for sub, name in (("depth_npy", "frame_00005.npy"), ("z_0", "frame_00005.npy"), ("mask_npy", "frame_00005.npy"), ("unedited", "frame_00005.jpg")):
        assert (tmp_path / sub / name).exists()
    # and how gc_dataset.py:52-66 reads them
    d = np.load(tmp_path / "depth_npy" / "frame_00005.npy")
    assert d.shape == (H, W, 1) and d.dtype == np.float32 and np.array_equal(d[:, :, 0], depth.numpy())
    z = np.load(tmp_path / "z_0" / "frame_00005.npy")
    assert z.shape == (1, 4, 8, 8) and np.array_equal(z, z0.numpy())
    assert midcache.has_view(tmp_path, 4) and not midcache.has_view(tmp_path, 5)
    back = midcache.load_view(tmp_path, 4)
    assert back["depth_image"].shape == (1, H, W) and torch.equal(back["depth_image"][0], depth)
    assert torch.equal(back["z_0_image"], z0) and torch.equal(back["mask_image"], mask)
    assert back["unedited_image"].shape == (H, W, 3) and float((back["unedited_image"] - rgb).abs().mean()) < 0.02   # JPEG


def test_checkpoint_layout(tmp_path):
    N = 10
    P = {"means": torch.randn(N, 3), "scales": torch.randn(N, 3), "quats": torch.randn(N, 4), "features_dc": torch.randn(N, 3),
         "features_rest": torch.randn(N, 15, 3), "opacities": torch.randn(N, 1)}
    path = tmp_path / "nerfstudio_models" / "step-000000499.ckpt"
    midcache.save_checkpoint(path, 499, P)
    raw = torch.load(path, weights_only=False)
    assert raw["step"] == 499 and set(raw["pipeline"]) == {f"_model.{k}" for k in midcache.CKPT_KEYS}
    step, back = midcache.load_checkpoint(path)
    assert step == 499 and all(torch.equal(back[k], P[k]) for k in P)
