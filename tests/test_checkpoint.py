"""gaussctrl_amd/sd/checkpoint.py -- the loader that replaces the reference's diffusers `from_pretrained` calls
(/root/reference/gaussctrl/gc_pipeline.py:97-102) -- executed on a diffusers-LAYOUT directory written to tmp_path: seeded random tensors of the
exact SD1.5 / sd-controlnet-depth / VAE inventories (gaussctrl_amd.sd.arch) as fp16 `diffusion_pytorch_model.safetensors` files in `unet/`,
`vae/` and a ControlNet folder.  CPU only (no kernels): what is tested is file discovery, key filtering, the pre-0.15 VAE attention key names
(query / key / value / proj_attn, 1x1-conv shaped), shape validation and the error behaviour (a missing file raises, nothing falls back to
random weights)."""
import os

import pytest
import torch

from gaussctrl_amd.sd import arch, checkpoint


def _rand(shapes, seed, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return {k: (torch.randn(*s, generator=g) * 0.02).to(dtype) for k, s in shapes.items()}


def _write(folder, sd, name="diffusion_pytorch_model.safetensors"):
    from safetensors.torch import save_file
    os.makedirs(folder, exist_ok=True)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(folder, name))


@pytest.fixture(scope="module")
def ckpt_dirs(tmp_path_factory):
    root = tmp_path_factory.mktemp("sd15")
    unet = _rand(arch.unet_shapes(), 1)
    unet["some.unused.buffer"] = torch.zeros(3, dtype=torch.float16)              # extra keys of a real file are ignored
    vae = {**_rand(arch.vae_decoder_shapes(), 2), **_rand(arch.vae_encoder_shapes(), 3)}
    # an old-style VAE file: attention linears named query / key / value / proj_attn and stored as 1x1 convolutions
    old = {}
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    for k, v in vae.items():
        if ".attentions." in k:
            for new, o in ren.items():
                if f".{new}." in k:
                    k = k.replace(f".{new}.", f".{o}.")
                    if v.dim() == 2:
                        v = v[:, :, None, None]
        old[k] = v
    cn = _rand(arch.controlnet_shapes(), 4)
    _write(os.path.join(root, "unet"), unet)
    _write(os.path.join(root, "vae"), old)
    cdir = tmp_path_factory.mktemp("controlnet_depth")
    _write(str(cdir), cn)
    return str(root), str(cdir), unet, vae, cn


def test_load_diffusion_weights_roundtrip(ckpt_dirs):
    root, cdir, unet, vae, cn = ckpt_dirs
    out = checkpoint.load_diffusion_weights(root, cdir)
    assert set(out) == {"unet", "controlnet", "vae_decoder", "vae_encoder"}
    assert set(out["unet"]) == set(arch.unet_shapes()) and "some.unused.buffer" not in out["unet"]
    assert set(out["controlnet"]) == set(arch.controlnet_shapes())
    assert set(out["vae_decoder"]) == set(arch.vae_decoder_shapes()) and set(out["vae_encoder"]) == set(arch.vae_encoder_shapes())
    for k in ("conv_in.weight", "mid_block.attentions.0.transformer_blocks.0.attn1.to_q.weight", "up_blocks.3.resnets.2.conv2.bias"):
        assert torch.equal(out["unet"][k], unet[k])
    for k, v in out["controlnet"].items():
        assert torch.equal(v, cn[k])
    # the renamed / reshaped VAE attention tensors came back under today's names with today's shapes and the same values
    for part in ("vae_decoder", "vae_encoder"):
        att = [k for k in out[part] if ".attentions." in k]
        assert att
        for k in att:
            assert out[part][k].shape == vae[k].shape and torch.equal(out[part][k], vae[k])


def test_controlnet_subfolder_and_bin(ckpt_dirs, tmp_path):
    """a ControlNet stored under <dir>/controlnet/ and as a `.bin` pickle of tensors is found as well"""
    root, _, _, _, cn = ckpt_dirs
    sub = tmp_path / "cn_repo" / "controlnet"
    os.makedirs(sub)
    torch.save(cn, sub / "diffusion_pytorch_model.bin")
    out = checkpoint.load_diffusion_weights(root, str(tmp_path / "cn_repo"))
    k = next(iter(arch.controlnet_shapes()))
    assert torch.equal(out["controlnet"][k], cn[k])


def test_missing_and_malformed_raise(ckpt_dirs, tmp_path, monkeypatch):
    root, cdir, unet, _, _ = ckpt_dirs
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    with pytest.raises(FileNotFoundError):
        checkpoint.load_diffusion_weights(str(tmp_path / "nowhere"), cdir)                 # not a directory, not in the hub cache
    empty = tmp_path / "empty"
    os.makedirs(empty / "unet")
    with pytest.raises(FileNotFoundError):
        checkpoint.load_diffusion_weights(str(empty), cdir)                                 # no diffusion_pytorch_model.* under unet/
    # a UNet with one tensor of the wrong shape and one missing key is rejected by the shape check, not silently accepted
    bad = dict(unet); bad.pop("some.unused.buffer")
    bad["conv_in.weight"] = torch.zeros(320, 4, 1, 1, dtype=torch.float16)
    bad.pop("conv_out.bias")
    broot = tmp_path / "bad"
    _write(str(broot / "unet"), bad)
    os.symlink(os.path.join(root, "vae"), broot / "vae")
    with pytest.raises(Exception) as ei:
        checkpoint.load_diffusion_weights(str(broot), cdir)
    assert "conv_in.weight" in str(ei.value) or "conv_out.bias" in str(ei.value)


def test_loaded_weights_prepare_like_synthetic(ckpt_dirs):
    """the loaded state dict goes through the same weights.prepare() the synthetic weights do (host-side re-layout only; CPU tensors)"""
    from gaussctrl_amd.sd.weights import prepare
    root, cdir, _, _, _ = ckpt_dirs
    out = checkpoint.load_diffusion_weights(root, cdir)
    w = prepare(out["controlnet"], torch.bfloat16, "cpu", heads=8)
    assert w["conv_in.weight"].dtype == torch.bfloat16 and w.get("_attn_q_prescaled")
