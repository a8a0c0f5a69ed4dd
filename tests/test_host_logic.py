"""CPU tests of the host side: C-ABI exports, constructor/state-dict parity with the reference, schedule mirrors,
weight packing.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import load_golden, ROOT


def test_library_loads_and_exports_every_declared_symbol():
    from anyedit_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "anyedit_hip.h")).read()
    declared = set(re.findall(r"\b(ae_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/anyedit_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.lib.ae_version() >= 100


def test_error_reporting_without_gpu():
    from anyedit_amd import _lib
    rc = _lib.lib.ae_gemm_bf16(None, 0, None, 0, 0, None, 0, None, 0, 1, 4, 64, None, None, 0, None, 0, 0, 0, 0, None, None)
    assert rc == -1 and b"null pointer" in _lib.lib.ae_last_error()
    rc = _lib.lib.ae_ddim_step_f32(1, 1, None, 1, None, None, 10, 7, 0, 0, 0, 0, 0, 0, 0, 0, None)
    assert rc == -1 and b"branches" in _lib.lib.ae_last_error()
    with pytest.raises(_lib.AnyEditHipError):
        _lib.check(rc, "ae_ddim_step_f32")


def test_ops_refuse_cpu_tensors():
    from anyedit_amd import ops
    with pytest.raises(ValueError, match="no CPU path"):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(4, 64, dtype=torch.bfloat16))


def test_constructor_matches_reference_init_and_state_dict_keys():
    """Same seed -> same parameters as the reference constructor (creation order preserved), same key schema."""
    from util_models import build_tiny_unet
    g = load_golden("unet_tiny")
    unet = build_tiny_unet()
    sd = unet.state_dict()
    ref_keys = {k[2:] for k in g if k.startswith("w.")}
    assert set(sd.keys()) == ref_keys
    for k in ref_keys:
        assert torch.equal(sd[k], torch.from_numpy(g["w." + k])), k


def test_guided_diffusion_options_key_schema():
    """resblock_updown / use_scale_shift_norm / conv_resample=False (openaimodel.py:178-274, 600-616, 707-721): the mirror's key set and shapes are the reference's
    (tests/golden/unet_gd_tiny.npz holds the reference constructor's state dicts)."""
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel, ResBlock
    from test_oracle_golden import GD_TINY
    g = load_golden("unet_gd_tiny")
    from test_oracle_golden import GD_ADM
    for tag, extra in dict({"updown_ssn": dict(resblock_updown=True, use_scale_shift_norm=True), "noconv": dict(conv_resample=False)}, **GD_ADM).items():
        sd = UNetModel(**dict(GD_TINY, **extra)).state_dict()
        ref = {k[len(tag) + 3:]: v.shape for k, v in g.items() if k.startswith(tag + ".w.")}
        assert set(sd.keys()) == set(ref.keys()), tag
        assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref), tag
    rb = ResBlock(64, 128, 0.0, out_channels=96, use_scale_shift_norm=True, down=True)
    assert rb.emb_layers[1].weight.shape == (192, 128) and rb.updown and not any(True for _ in rb.h_upd.parameters())
    cb = UNetModel(**dict(GD_TINY, conv_resample=False, n_embed=24)).state_dict()                 # predict_codebook_ids: `out` stays, id_predictor is added
    ref = {k[len("noconv.w."):] for k in g if k.startswith("noconv.w.")} | {k[len("codebook.w."):] for k in g if k.startswith("codebook.w.")}
    assert set(cb.keys()) == ref and tuple(cb["id_predictor.1.weight"].shape) == (24, 32, 1, 1)
    with pytest.raises(NotImplementedError):
        UNetModel(**dict(GD_TINY, dims=3))
    with pytest.raises(AssertionError):
        UNetModel(**dict(GD_TINY, use_spatial_transformer=False))        # a context_dim without the spatial transformer (openaimodel.py:477-478)
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import AttentionBlock, QKVAttention, QKVAttentionLegacy
    heads = lambda m: [b.num_heads for b in m.modules() if isinstance(b, AttentionBlock)]   # noqa: E731
    assert heads(UNetModel(**dict(GD_TINY, **GD_ADM["adm"]))) == [2, 4, 4, 4, 4, 2, 2]         # ch // 16 at 32 / 64 channels
    assert heads(UNetModel(**dict(GD_TINY, **GD_ADM["adm_legacy"]))) == [2] * 7
    assert heads(UNetModel(**dict(GD_TINY, use_spatial_transformer=False, context_dim=None, num_heads=4, num_heads_upsample=2, legacy=True))) == [4, 4, 4, 2, 2, 2, 2]
    assert isinstance(AttentionBlock(64, num_head_channels=16, use_new_attention_order=True).attention, QKVAttention)
    assert isinstance(AttentionBlock(64, num_heads=4).attention, QKVAttentionLegacy)


def test_sd15_key_schema_and_param_count():
    """Appendix A of SURVEY.md: 686 tensors, 859 532 484 parameters (meta device, no memory)."""
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    with torch.device("meta"):
        unet = UNetModel(image_size=64, in_channels=8, model_channels=320, out_channels=4, num_res_blocks=2,
                         attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_heads=8,
                         use_spatial_transformer=True, transformer_depth=1, context_dim=768, legacy=False)
    sd = unet.state_dict()
    assert len(sd) == 686
    assert sum(v.numel() for v in sd.values()) == 859532484
    assert sd["input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight"].shape == (320, 768)
    assert sd["output_blocks.5.2.conv.weight"].shape == (1280, 1280, 3, 3)
    assert sd["input_blocks.3.0.op.weight"].shape == (320, 320, 3, 3)
    assert "output_blocks.11.0.skip_connection.weight" in sd and "input_blocks.1.0.skip_connection.weight" not in sd


def test_schedule_mirror_bit_exact():
    from anyedit_amd.ldm.modules.diffusionmodules import util as U
    g = load_golden("schedule")
    assert np.array_equal(U.make_beta_schedule("linear", 1000, 0.00085, 0.0120), g["betas"])
    for s in (7, 20, 30, 50, 100):
        ts = U.make_ddim_timesteps("uniform", s, 1000, verbose=False)
        assert ts.dtype == g[f"ts_uniform_{s}"].dtype and np.array_equal(ts, g[f"ts_uniform_{s}"])
    assert np.array_equal(U.make_ddim_timesteps("quad", 10, 1000, verbose=False), g["ts_quad_10"])
    ac = np.cumprod(1.0 - g["betas"], axis=0)
    sig, a, ap = U.make_ddim_sampling_parameters(ac, g["ts_uniform_50"], 1.0, verbose=False)
    assert np.array_equal(sig, g["sig_S50_eta1"]) and np.array_equal(a, g["a_S50_eta1"]) and np.array_equal(ap, g["ap_S50_eta1"])


def test_ddpm_buffers_and_sampler_schedule_on_cpu():
    """DDPM.register_schedule + DDIMSampler.make_schedule (host part) against the reference's tables, bit-exact."""
    from anyedit_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler
    from util_models import build_tiny_unet
    g = load_golden("ddim_tiny")
    ldm = LatentDiffusion(build_tiny_unet(), conditioning_key="hybrid", timesteps=1000, linear_start=0.00085, linear_end=0.0120)
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
        assert torch.equal(getattr(ldm, k), torch.from_numpy(g[f"model.{k}"])), k
    s = DDIMSampler(ldm)
    for tag, S in (("s5_cfg", 5), ("s20_cfg", 20), ("s7_cfg_mask", 7)):
        s.make_schedule(S, ddim_eta=0.0, verbose=False)
        assert np.array_equal(s.ddim_timesteps, g[f"{tag}.ddim_timesteps"]) and s.ddim_timesteps.dtype == np.int64
        for k in ("ddim_alphas", "ddim_alphas_prev", "ddim_sigmas", "ddim_sqrt_one_minus_alphas"):
            assert np.array_equal(np.asarray(getattr(s, k)), g[f"{tag}.{k}"]), k
        # the fp32 scalars handed to the kernel are exactly what torch.full would have stored
        for idx in range(len(s.ddim_timesteps)):
            s1m, sat, sap, dirc, sig = s._coeffs(idx, False)
            a_t = torch.full((1,), g[f"{tag}.ddim_alphas"][idx])
            a_prev = torch.full((1,), g[f"{tag}.ddim_alphas_prev"][idx])
            sg = torch.full((1,), g[f"{tag}.ddim_sigmas"][idx])
            # torch's CPU fp32 sqrt is not always correctly rounded (1 ulp off for some inputs); numpy's is,
            # like the GPU's sqrt the reference would run -> compare within 1 ulp
            ulp = lambda a, b: abs(np.float32(a).view(np.int32).astype(np.int64) - np.float32(b).view(np.int32).astype(np.int64)) <= 1
            assert ulp(sat, float(a_t.sqrt())) and ulp(sap, float(a_prev.sqrt()))
            assert ulp(dirc, float((1. - a_prev - sg ** 2).sqrt()))
            assert s1m == float(torch.full((1,), g[f"{tag}.ddim_sqrt_one_minus_alphas"][idx]))


def test_weight_packing_layouts():
    from anyedit_amd import ops
    w = (torch.arange(2 * 3 * 9, dtype=torch.float32) % 97).reshape(2, 3, 3, 3)
    p = ops.pack_conv3x3(w).float().reshape(2, 9, 64)
    for ky in range(3):
        for kx in range(3):
            assert torch.equal(p[:, ky * 3 + kx, :3], w[:, :, ky, kx]) and p[:, ky * 3 + kx, 3:].abs().sum() == 0
    # chunk-major K order (k_order = 1): column (c // 64) * 9 * 64 + tap * 64 + c % 64 holds w[:, c, ky, kx] — the same values, re-ordered
    w2 = (torch.arange(2 * 128 * 9, dtype=torch.float32) % 251).reshape(2, 128, 3, 3)
    p0, p1 = ops.pack_conv3x3(w2).float(), ops.pack_conv3x3(w2, k_order=1).float()
    for c in (0, 5, 63, 64, 100, 127):
        for tap in range(9):
            assert torch.equal(p1[:, (c // 64) * 576 + tap * 64 + c % 64], w2[:, c, tap // 3, tap % 3])
            assert torch.equal(p0[:, tap * 128 + c], w2[:, c, tap // 3, tap % 3])
    assert torch.equal(p0.sort(dim=1).values, p1.sort(dim=1).values)
    inner = 32
    wg = (torch.arange(2 * inner * 8, dtype=torch.float32) % 101).reshape(2 * inner, 8)
    bg = torch.arange(2 * inner, dtype=torch.float32)
    wp, bp = ops.pack_geglu(wg, bg)
    wp = wp.float()
    for j in range(inner // 16):
        assert torch.equal(wp[32 * j:32 * j + 16], wg[16 * j:16 * j + 16])
        assert torch.equal(wp[32 * j + 16:32 * j + 32], wg[inner + 16 * j:inner + 16 * j + 16])
        assert torch.equal(bp[32 * j + 16:32 * j + 32], bg[inner + 16 * j:inner + 16 * j + 16])
    # the stem conv's tap-packed weight (im2col + dense GEMM): column 8 (3 ky + kx) + cin, zeros from column 72 on
    ws = (torch.arange(5 * 8 * 9, dtype=torch.float32) % 61).reshape(5, 8, 3, 3)
    wt = ops.pack_conv3x3_taps8(ws).float()
    assert wt.shape == (5, 128) and float(wt[:, 72:].abs().max()) == 0.0
    for tap in range(9):
        assert torch.equal(wt[:, 8 * tap:8 * tap + 8], ws[:, :, tap // 3, tap % 3])


def test_sam_key_schema_and_prompt_geometry():
    """N3: the Sam mirror has the reference's state-dict schema (golden keys of the prompt encoder / mask decoder; build_sam_vit_b's
    93.7 M parameters), ResizeLongestSide reproduces the reference's arithmetic, SamPredictor keeps its error behaviour."""
    import numpy as np
    from anyedit_amd.segment_anything import SamPredictor, sam_model_registry
    from anyedit_amd.segment_anything.modeling import MaskDecoder, PromptEncoder, TwoWayTransformer
    from anyedit_amd.segment_anything.utils.transforms import ResizeLongestSide
    g = load_golden("sam_decoder")
    pe = PromptEncoder(embed_dim=64, image_embedding_size=(8, 8), input_image_size=(128, 128), mask_in_chans=16)
    md = MaskDecoder(num_multimask_outputs=3, transformer=TwoWayTransformer(depth=2, embedding_dim=64, mlp_dim=128, num_heads=4),
                     transformer_dim=64, iou_head_depth=3, iou_head_hidden_dim=64)
    keys = {"prompt_encoder." + k for k in pe.state_dict()} | {"mask_decoder." + k for k in md.state_dict()}
    assert keys == {k[2:] for k in g if k.startswith("w.")}
    for k, v in pe.state_dict().items():
        assert tuple(v.shape) == g["w.prompt_encoder." + k].shape, k
    for k, v in md.state_dict().items():
        assert tuple(v.shape) == g["w.mask_decoder." + k].shape, k
    with torch.device("meta"):
        sam = sam_model_registry["vit_b"]()
    assert sum(p.numel() for p in sam.parameters()) == 93_735_472
    assert "pixel_mean" not in sam.state_dict()                                    # non-persistent buffers (sam.py:49-50)
    t = ResizeLongestSide(1024)
    assert t.get_preprocess_shape(512, 768, 1024) == (683, 1024) and t.get_preprocess_shape(75, 100, 128) == (96, 128)
    b = t.apply_boxes_torch(torch.tensor([[10.0, 20.0, 300.0, 400.0]]), (512, 768))
    assert torch.allclose(b, torch.tensor([[10 * 1024 / 768, 20 * 683 / 512, 300 * 1024 / 768, 400 * 683 / 512]]))
    assert np.allclose(t.apply_boxes(np.array([[10.0, 20.0, 300.0, 400.0]]), (512, 768)), b.numpy())
    img = (np.arange(30 * 40 * 3) % 251).astype(np.uint8).reshape(30, 40, 3)
    assert ResizeLongestSide(128).apply_image(img).shape == (96, 128, 3)
    pred = SamPredictor.__new__(SamPredictor)
    pred.reset_image()
    with pytest.raises(RuntimeError, match="set_image"):
        pred.get_image_embedding()


SD15_UNET = dict(image_size=64, in_channels=8, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                 channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768, legacy=False)


def test_checkpoint_layouts_and_diffusers_key_mapping(tmp_path):
    """N4 on-disk formats: ldm <-> diffusers key layouts (derived from the module structure), the three file forms a UNet arrives in,
    and cldm.model's checkpoint readers."""
    import safetensors.torch
    from util_models import build_tiny_unet
    from anyedit_amd import checkpoints as C
    from anyedit_amd.cldm.model import get_state_dict, load_state_dict
    from anyedit_amd.ldm.models.autoencoder import AutoencoderKL
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    with torch.device("meta"):
        unet = UNetModel(**SD15_UNET)
    km = C.unet_ldm_to_diffusers_keys(unet)
    assert len(km) == 686 and len(set(km.values())) == 686
    known = {"input_blocks.0.0.weight": "conv_in.weight", "time_embed.0.weight": "time_embedding.linear_1.weight",
             "input_blocks.1.0.in_layers.2.weight": "down_blocks.0.resnets.0.conv1.weight",
             "input_blocks.2.1.transformer_blocks.0.attn2.to_k.weight": "down_blocks.0.attentions.1.transformer_blocks.0.attn2.to_k.weight",
             "input_blocks.3.0.op.weight": "down_blocks.0.downsamplers.0.conv.weight",
             "input_blocks.4.0.skip_connection.weight": "down_blocks.1.resnets.0.conv_shortcut.weight",
             "input_blocks.11.0.emb_layers.1.bias": "down_blocks.3.resnets.1.time_emb_proj.bias",
             "middle_block.1.transformer_blocks.0.ff.net.0.proj.weight": "mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
             "middle_block.2.out_layers.3.weight": "mid_block.resnets.1.conv2.weight",
             "output_blocks.2.1.conv.weight": "up_blocks.0.upsamplers.0.conv.weight",
             "output_blocks.5.2.conv.weight": "up_blocks.1.upsamplers.0.conv.weight",
             "output_blocks.3.1.proj_out.bias": "up_blocks.1.attentions.0.proj_out.bias",
             "output_blocks.11.0.skip_connection.weight": "up_blocks.3.resnets.2.conv_shortcut.weight",
             "out.0.weight": "conv_norm_out.weight", "out.2.bias": "conv_out.bias"}
    for k, d in known.items():
        assert km[k] == d, (k, km[k])
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
              attn_resolutions=[], dropout=0.0)
    with torch.device("meta"):
        vae = AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4)
    vm = C.vae_ldm_to_diffusers_keys(vae)
    assert len(vm) == 248 and len(set(vm.values())) == 248
    assert vm["decoder.up.3.block.0.norm1.weight"] == "decoder.up_blocks.0.resnets.0.norm1.weight"
    assert vm["decoder.up.1.upsample.conv.weight"] == "decoder.up_blocks.2.upsamplers.0.conv.weight"
    assert vm["decoder.up.1.block.0.nin_shortcut.weight"] == "decoder.up_blocks.2.resnets.0.conv_shortcut.weight"
    assert vm["encoder.mid.attn_1.q.weight"] == "encoder.mid_block.attentions.0.to_q.weight"
    assert vm["encoder.down.2.downsample.conv.bias"] == "encoder.down_blocks.2.downsamplers.0.conv.bias"

    tiny = build_tiny_unet()
    sd = {k: v.clone() for k, v in tiny.state_dict().items()}
    dif = C.convert_unet_to_diffusers(tiny)
    assert set(dif) == set(C.unet_ldm_to_diffusers_keys(tiny).values())
    k1 = next(k for k in dif if k.endswith("attentions.0.proj_in.weight"))
    dif_lin = dict(dif)
    dif_lin[k1] = dif[k1].reshape(dif[k1].shape[0], -1)                       # the Linear flavour of the 1x1 projection
    back = C.convert_diffusers_unet(tiny, dif_lin)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    with pytest.raises(KeyError):
        C.convert_diffusers_unet(tiny, {k: v for k, v in dif.items() if k != "conv_in.weight"})
    with pytest.raises(ValueError):
        C.convert_diffusers_unet(tiny, {**dif, "conv_in.weight": dif["conv_in.weight"][:, :4]})

    files = {"ldm": (tmp_path / "unet_ldm.safetensors", sd), "diffusers": (tmp_path / "diffusion_pytorch_model.safetensors", dif),
             "ldm-checkpoint": (tmp_path / "sd.ckpt", None)}
    safetensors.torch.save_file({k: v.contiguous() for k, v in sd.items()}, str(files["ldm"][0]))
    safetensors.torch.save_file({k: v.contiguous() for k, v in dif.items()}, str(files["diffusers"][0]))
    full = {"model.diffusion_model." + k: v for k, v in sd.items()}
    full["first_stage_model.encoder.conv_in.weight"] = torch.zeros(1)
    torch.save({"state_dict": full, "global_step": 7}, str(files["ldm-checkpoint"][0]))
    for layout, (path, _) in files.items():
        fresh = build_tiny_unet()
        with torch.no_grad():
            for p in fresh.parameters():
                p.zero_()
        assert C.load_unet_weights(fresh, str(path)) == layout
        assert all(torch.equal(v, sd[k]) for k, v in fresh.state_dict().items()), layout
    assert get_state_dict({"state_dict": {"a": 1}}) == {"a": 1} and get_state_dict({"a": 1}) == {"a": 1}
    assert set(load_state_dict(str(files["ldm-checkpoint"][0]))) == set(full)


def test_create_model_from_reference_style_yaml(tmp_path):
    """cldm.model.create_model on a YAML whose `target:` strings name the reference's classes (anydoor.yaml:2,22,40,56), and the
    AnyDoor geometry (SD-2.1 widths: 64-wide heads, linear transformer projections, 1024-wide context) on the meta device."""
    from anyedit_amd.cldm.cldm import ControlLDM, ControlNet, ControlledUnetModel
    from anyedit_amd.cldm.model import create_model
    tiny = "image_size: 8\n        in_channels: 4\n        model_channels: 32\n        attention_resolutions: [1, 2]\n        " \
           "num_res_blocks: 1\n        channel_mult: [1, 2]\n        num_head_channels: 8\n        use_spatial_transformer: true\n        " \
           "use_linear_in_transformer: true\n        transformer_depth: 1\n        context_dim: 16\n        legacy: false\n"
    yaml_text = f"""model:
  target: AnyEdit_Collection.other_modules.cldm.cldm.ControlLDM
  params:
    linear_start: 0.00085
    linear_end: 0.0120
    timesteps: 1000
    image_size: 8
    channels: 4
    conditioning_key: crossattn
    scale_factor: 0.18215
    use_ema: false
    only_mid_control: false
    control_key: hint
    control_stage_config:
      target: AnyEdit_Collection.other_modules.cldm.cldm.ControlNet
      params:
        hint_channels: 4
        {tiny}
    unet_config:
      target: AnyEdit_Collection.other_modules.cldm.cldm.ControlledUnetModel
      params:
        out_channels: 4
        {tiny}
    first_stage_config:
      target: ldm.models.autoencoder.AutoencoderKL
      params:
        embed_dim: 4
        ddconfig: {{double_z: true, z_channels: 4, resolution: 32, in_channels: 3, out_ch: 3, ch: 32, ch_mult: [1, 2], num_res_blocks: 1,
                   attn_resolutions: [], dropout: 0.0}}
        lossconfig:
          target: torch.nn.Identity
"""
    path = tmp_path / "tiny_anydoor.yaml"
    path.write_text(yaml_text)
    model = create_model(str(path))
    assert isinstance(model, ControlLDM) and isinstance(model.control_model, ControlNet)
    assert isinstance(model.model.diffusion_model, ControlledUnetModel) and model.first_stage_model is not None
    assert model.scale_factor == 0.18215 and float(model.betas[0]) == pytest.approx(0.00085, rel=1e-6)
    geo = dict(image_size=32, in_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
               channel_mult=[1, 2, 4, 4], num_head_channels=64, use_spatial_transformer=True, use_linear_in_transformer=True,
               transformer_depth=1, context_dim=1024, legacy=False)
    with torch.device("meta"):
        unet = ControlledUnetModel(out_channels=4, **geo)
        cnet = ControlNet(hint_channels=4, **geo)
    assert (sum(p.numel() for p in unet.parameters()), len(unet.state_dict())) == (865_910_724, 686)      # [probe] of the reference classes
    assert (sum(p.numel() for p in cnet.parameters()), len(cnet.state_dict())) == (364_228_384, 340)


def test_python_plan_mirror_matches_the_library():
    """ops._conv_splitk (used only to LABEL profiler rows with the kernel that will run) must agree with the launch plan inside the
    library; ae_conv3x3_workspace_floats is host-only and returns splitk * M * Cout (0 when unsplit), so it can be checked without a GPU."""
    from anyedit_amd import ops
    from anyedit_amd._lib import lib
    for B in (1, 3, 4, 12):
        for hw in (8, 16, 32, 64):
            for cin, cout in ((320, 320), (640, 640), (1280, 1280), (2560, 1280), (1920, 640), (960, 320), (640, 1280), (8, 320), (320, 4)):
                M = B * hw * hw
                ws = lib.ae_conv3x3_workspace_floats(B, hw, hw, cin, cout, 1, 0)
                s_lib = ws // (M * cout) if ws else 1
                cin_pad = (cin + 63) // 64 * 64
                assert ops._conv_splitk(M, cout, 9 * cin_pad) == s_lib, (B, hw, cin, cout, s_lib)


def test_layernorm_fold_plan_follows_the_tile_plan(monkeypatch):
    """ops.ln_fold_plan / ae_gemm_ln_plan (host-only: the library runs its launch selection without launching): at the bench's UNet batch 12
    every norm -> projection pair of the 32x32 and 16x16 levels folds (producer emits row statistics, consumer applies them); since round 5 the
    64x64 level folds too — on the row-panel kernel's own fold forms (AE_RP_FOLD=0 restores its LayerNorm prologue: then the plan says no there);
    the 8x8 level and shapes whose producer runs another tile fall back; nothing folds while the training tape records or under AE_LN_FOLD=0."""
    from anyedit_amd import ops
    from anyedit_amd._lib import lib
    if not ops._LN_FOLD:
        pytest.skip("AE_LN_FOLD=0")
    E, G = ops.EPI_NONE, ops.EPI_GEGLU
    for M, C in ((12 * 1024, 640), (12 * 256, 1280)):
        assert ops.ln_fold_plan(M, C, C, E, 1), (M, C)                                     # proj_in / to_out emit
        assert ops.ln_fold_plan(M, 3 * C, C, E, 2) and ops.ln_fold_plan(M, C, C, E, 2)     # qkv, q
        assert ops.ln_fold_plan(M, 8 * C, C, G, 2)                                         # GEGLU projection
    M = 12 * 4096
    assert lib.ae_ln_gemm_supported(M, 960, 320, E) == 1
    if __import__("os").environ.get("AE_RP_FOLD", "1") != "0":
        assert ops.ln_fold_plan(M, 960, 320, E, 2) and ops.ln_fold_plan(M, 320, 320, E, 2) and ops.ln_fold_plan(M, 320, 320, E, 1) and ops.ln_fold_plan(M, 2560, 320, G, 2)
        assert lib.ae_gemm_ln_plan(M, 2560, 320, G, 1) == 0                                # statistics go with a plain GEMM (+bias, +residual), not with GEGLU
    else:
        assert not ops.ln_fold_plan(M, 960, 320, E, 2) and not ops.ln_fold_plan(M, 320, 320, E, 1) and not ops.ln_fold_plan(M, 2560, 320, G, 2)
    assert not ops.ln_fold_plan(12 * 64, 1280, 1280, E, 1)                                 # 8x8 level: 64x64 tiles, no statistics epilogue
    assert lib.ae_gemm_ln_plan(M, 640, 640, G, 1) == 0 and lib.ae_gemm_ln_plan(M, 600, 640, E, 1) == 0 and lib.ae_gemm_ln_plan(M, 640, 640, E, 3) == 0
    # a plan answer is a property of (M, N, K, epilogue): the same question twice, and the launch-side refusal for an uncovered shape
    assert lib.ae_gemm_ln_plan(3072, 1280, 1280, E, 1) == lib.ae_gemm_ln_plan(3072, 1280, 1280, E, 1) == 1

    # ... and the launch side refuses such a shape BEFORE anything is launched (the selection runs once as a query): no GPU is touched here
    import ctypes
    buf = (ctypes.c_char * 4096)()
    ptr = (ctypes.addressof(buf) + 255) // 256 * 256
    rc = lib.ae_gemm_ln_bf16(ptr, 640, ptr, 640, ptr, 640, 64, 640, 640, ptr, None, 0, 0, ptr, None, 0, None, 0.0, None)
    assert rc == -3 and b"no row-statistics instantiation" in lib.ae_last_error()
    rc = lib.ae_gemm_ln_bf16(ptr, 640, ptr, 640, ptr, 640, 64, 640, 640, ptr, None, 0, 0, ptr, ptr, 10, ptr, 1e-5, None)
    assert rc == -1 and b"exactly one of" in lib.ae_last_error()

    class _Tape:
        active = True
    monkeypatch.setattr(ops, "_TAPE", _Tape())
    assert not ops.ln_fold_plan(12 * 1024, 640, 640, E, 1)
    monkeypatch.setattr(ops, "_TAPE", None)
    monkeypatch.setattr(ops, "_LN_FOLD", False)
    assert not ops.ln_fold_plan(12 * 1024, 640, 640, E, 1)


@pytest.mark.parametrize("B,H,W,Cin", [(2, 64, 64, 128), (12, 16, 16, 64), (3, 32, 32, 64), (1, 96, 96, 64)])
def test_slab_layout_of_the_conv_loop_restated(B, H, W, Cin):
    """The address arithmetic of gemm_conv.hip's slab loop (WA = 4), restated lane by lane in numpy: the LDS-DMA pieces of one (chunk, ky) slab
    — zero row in front of every image row, rows outside the image zero, XOR swizzle on the source chunk — and the fragment reads of the three
    kx taps, against the definition of a zero-padded 3x3 gather.  Covers tiles that cross a sample boundary, the ragged last tile and image
    widths 16 .. 96 (the kernel's preconditions: 16 | W, W | 192).  No GPU: this pins the LAYOUT; tests/test_hip_* pin the kernel."""
    import numpy as np
    BM, NWAVE = 192, 8
    SLAB_P = (BM + BM // 16 + 1 + 7) // 8
    M = B * H * W
    rng = np.random.default_rng(B + H + Cin)
    x = rng.integers(1, 60000, size=(M, Cin), dtype=np.uint16)   # non-zero bf16 bit patterns
    xb = x.view(np.uint8).reshape(-1)
    W1, nrow, ntm = W + 1, BM // W, (M + BM - 1) // BM
    assert W % 16 == 0 and BM % W == 0 and BM + nrow + 1 <= SLAB_P * 8
    for tile in sorted({0, 1, ntm // 2, ntm - 1}):
        m0 = tile * BM
        y0 = (m0 // W) % H
        c = (Cin // 64) - 1
        for ky in range(3):
            lds = np.full(SLAB_P * 1024, 0xAB, dtype=np.uint8)
            for wave in range(NWAVE):
                for j in range((SLAB_P + NWAVE - 1) // NWAVE):
                    q = wave + NWAVE * j
                    if q >= SLAB_P:
                        q -= NWAVE                                   # the wave repeats its previous piece
                    for lane in range(64):
                        srow, lc = 8 * q + (lane >> 3), (lane & 7) ^ (lane >> 3)
                        jr = srow // W1
                        pos = srow - jr * W1
                        m = m0 + jr * W + pos - 1
                        y = y0 + jr - (H if y0 + jr >= H else 0)
                        ok = pos > 0 and jr < nrow and m < M and 0 <= y + ky - 1 < H
                        voff = m * Cin * 2 + lc * 16 + (ky - 1) * W * Cin * 2
                        dst = q * 1024 + lane * 16
                        if ok and 0 <= voff and voff + 16 <= xb.size:            # the descriptor's bounds check is on voffset alone
                            lds[dst:dst + 16] = xb[voff + c * 128:voff + c * 128 + 16]
                        else:
                            lds[dst:dst + 16] = 0
            for wm in range(2):
                for i in range(6):
                    r0 = wm * 96 + i * 16
                    for l15 in range(16):
                        base, m = r0 + r0 // W + l15, m0 + r0 + l15
                        if m >= M:
                            continue
                        b, rem = divmod(m, H * W)
                        oy, ox = divmod(rem, W)
                        for kx in range(3):
                            iy, ix = oy + ky - 1, ox + kx - 1
                            for kk in range(2):
                                for lg in range(4):
                                    addr = (base + kx) * 128 + (((kk * 4 + lg) ^ ((base + kx) & 7)) << 4)
                                    exp = np.zeros(16, np.uint8)
                                    if 0 <= iy < H and 0 <= ix < W:
                                        ch = c * 64 + (kk * 4 + lg) * 8
                                        exp = x[(b * H + iy) * W + ix, ch:ch + 8].view(np.uint8)
                                    assert np.array_equal(lds[addr:addr + 16], exp), (tile, ky, kx, wm, i, l15, kk, lg)


def test_conv_k_order_follows_the_tile_plan(monkeypatch):
    """ops.conv_k_order (which weight pack / K order a conv launch is given).  Shipped default (AE_CONV_KMAJOR = 2, round 4): chunk-major where
    the un-split 192x320 plan runs — the 64x64-level convs of a UNet batch >= 12 — AND where its split-K form runs — the 16x16-level convs at
    batch 12 (M = 3072: 16 row tiles x 4 column tiles, K cut so that 256 blocks exist) —, tap-major for every other grid, for upsampling convs,
    for channel counts that are not multiples of 64 and under the training tape.  Knob 1 (round 3's rule) drops the split-K half, knob 0 is
    tap-major everywhere.  VERDICT r4: the test used to skip itself unless the knob was 1 — at the shipped default nothing checked the rule."""
    from anyedit_amd import ops
    monkeypatch.delenv("AE_GEMM_GLDS", raising=False)
    assert int(__import__("os").environ.get("AE_CONV_KMAJOR", "2")) == ops._CONV_KMAJOR
    for knob in (2, 1, 0):
        monkeypatch.setattr(ops, "_CONV_KMAJOR", knob)
        on = 1 if knob else 0
        for B, want in ((12, on), (24, on), (1, 0), (4, 0)):
            for cin in (320, 640, 960):
                assert ops.conv_k_order(B * 64 * 64, cin, 320) == want, (knob, B, cin)
        assert ops.conv_k_order(12 * 64 * 64, 640, 640, upsample2x=True) == 0          # the up-conv keeps the tap-major gather
        assert ops.conv_k_order(12 * 64 * 64, 8, 320) == 0 and ops.conv_k_order(12 * 64 * 64, 320, 4) == 0   # stem / head
        assert ops.conv_k_order(12 * 32 * 32, 640, 640) == 0 and ops.conv_k_order(12 * 8 * 8, 1280, 1280) == 0
        # the split-K rule: the 16x16 level at UNet batch 12 (every Cin of that level), not at batch 24 (128 tiles: un-split 128x128 plan) or batch 4
        sk = 1 if knob == 2 else 0
        for cin in (640, 1280, 1920, 2560):
            assert ops._tile_label(12 * 16 * 16, 1280, True, 9 * cin, False, True) == "192x320,splitK", cin
            assert ops.conv_k_order(12 * 16 * 16, cin, 1280) == sk, (knob, cin)
        assert ops.conv_k_order(12 * 16 * 16, 1280, 1280, stride=2) == sk               # (a stride-2 launch of that grid takes the same plan; its slab form is refused by the launcher, not here)
        assert ops.conv_k_order(4 * 16 * 16, 1280, 1280) == 0
        # the slab form of the loop needs every split-K block's K range to start at a chunk boundary (launcher: ceil(KT / split) % 9 == 0): the two
        # frequent convs of the level (Cin 1280: 6 launches per evaluation, 2560: 2) qualify, Cin 640 / 1920 (one launch each) run the tap form
        for cin, slab in ((640, False), (1280, True), (1920, False), (2560, True)):
            sp = ops._conv_t320_split(12 * 16 * 16, 1280, 9 * cin)
            kt = 9 * cin // 64
            assert sp == 4 and ((-(-kt // sp)) % 9 == 0) == slab, (cin, sp)
    # round 5: the split plan also takes the 32x32-level convs with >= 180 K tiles (AE_CONV_T320_SPLITK default 3), in the chunk-major order; not the shorter ones
    monkeypatch.setattr(ops, "_CONV_KMAJOR", 2)
    monkeypatch.delenv("AE_CONV_T320_SPLITK", raising=False)
    assert ops._conv_t320_split(12 * 32 * 32, 640, 9 * 1920) == 2 and ops._conv_t320_split(12 * 32 * 32, 640, 9 * 1280) == 2
    assert ops.conv_k_order(12 * 32 * 32, 1920, 640) == 1 and ops.conv_k_order(12 * 32 * 32, 1280, 640) == 1
    assert ops._conv_t320_split(12 * 32 * 32, 640, 9 * 960) == 0 and ops.conv_k_order(12 * 32 * 32, 960, 640) == 0 and ops.conv_k_order(12 * 32 * 32, 640, 640) == 0
    monkeypatch.setenv("AE_CONV_T320_SPLITK", "2")
    assert ops._conv_t320_split(12 * 32 * 32, 640, 9 * 1920) == 0 and ops.conv_k_order(12 * 32 * 32, 1920, 640) == 0
    monkeypatch.delenv("AE_CONV_T320_SPLITK", raising=False)
    monkeypatch.setenv("AE_GEMM_GLDS", "0")
    assert ops.conv_k_order(12 * 64 * 64, 320, 320) == 0                               # the chunk-major order exists in the LDS-DMA loader only


def test_upsample_conv_as_four_2x2_convs_identity():
    """Round 5 (openaimodel.py:108-118): conv3x3(nearest_upsample_x2(X), pad 1) restated as four 2x2 convs on X with summed taps — the identity behind
    `ops.pack_conv3x3_up2` / `ae_conv3x3_up2_bf16`, checked in fp32 on the CPU against torch's interpolate + conv2d, including the image border (the
    zero padding of the upsampled map is the zero padding of the low-resolution map) and the tap order / parity layout the kernel assumes:
    set p = 2 py + px, tap t = 2 i + j reads input pixel (y + py - 1 + i, x + px - 1 + j) and writes output pixel (2 y + py, 2 x + px)."""
    import torch.nn.functional as F
    from anyedit_amd import ops
    g = torch.Generator().manual_seed(31)
    B, C, Co, H, W = 2, 64, 8, 5, 7
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, padding=1)
    ws = ops.up2_weight_sums(w)                                  # [4, Co, 4, C]
    xp = F.pad(x, (1, 1, 1, 1))                                  # zero border of the LOW-resolution map
    out = torch.zeros(B, Co, 2 * H, 2 * W)
    for py in range(2):
        for px in range(2):
            acc = torch.zeros(B, Co, H, W)
            for i in range(2):
                for j in range(2):
                    win = xp[:, :, py + i:py + i + H, px + j:px + j + W]     # input pixel (y + py - 1 + i, x + px - 1 + j)
                    acc += torch.einsum("bchw,oc->bohw", win, ws[2 * py + px, :, 2 * i + j])
            out[:, :, py::2, px::2] = acc
    assert float((out - ref).abs().max()) < 2e-4 * float(ref.abs().max())
    packed = ops.pack_conv3x3_up2(w)
    assert tuple(packed.shape) == (4, Co, 4 * C) and packed.dtype == torch.bfloat16
    assert torch.equal(packed.reshape(4, Co, 4, C), ws.to(torch.bfloat16))


def test_fused_feed_forward_weight_image_index_map():
    """Round 6 (attention.py:49-76): `ops.pack_ff2_fused` against the data flow of csrc/ff_fused.hip restated with index arithmetic on the CPU.  The gated
    values of hidden chunks 2 s, 2 s + 1 leave P1 in lane (row n = lane & 15, g = lane >> 4) as elements e = 4 j + r <-> unit 32 s + 16 j + 4 g + r and are fed
    to the next MFMA as its B operand as they are; lane (i = lane & 15, g) reads 16 bytes of image row 16 cf + i at position (g + 2 (i >> 2)) & 3 as the A
    operand; D[i][n] = sum over (g, e); result row i = 4 g' + r of fragment cf is output column 32 (cf >> 1) + 8 g' + 4 (cf & 1) + r."""
    from anyedit_amd import ops
    g_ = torch.Generator().manual_seed(5)
    C, H, R = 320, 128, 16
    w2 = torch.randn(C, H, generator=g_)
    hval = torch.randn(R, H, generator=g_).to(torch.bfloat16).float()
    img = ops.pack_ff2_fused(w2).float()                       # [H / 32, C, 32]
    assert tuple(img.shape) == (H // 32, C, 32)
    out = torch.zeros(R, C)
    for s in range(H // 32):
        for cf in range(20):
            D = torch.zeros(16, R)
            for i in range(16):
                for gg in range(4):
                    pos = (gg + 2 * (i >> 2)) & 3
                    a = img[s, 16 * cf + i, 8 * pos:8 * pos + 8]                                               # lane (i, gg)'s A fragment
                    units = torch.tensor([32 * s + 16 * (e >> 2) + 4 * gg + (e & 3) for e in range(8)])
                    D[i] += hval[:, units] @ a                                                                 # B fragment of lane (n, gg) = hval[n, units]
            for gp in range(4):
                for r in range(4):
                    out[:, 32 * (cf >> 1) + 8 * gp + 4 * (cf & 1) + r] += D[4 * gp + r]
    ref = hval @ w2.to(torch.bfloat16).float().t()
    assert float((out - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # conflict-free: the 16 lanes of a ds_read_b128 group ({0-3, 12-15, 20-27}) touch 16 different 16-byte slots of the 256-byte bank row
    for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
        slots = {((lane & 15) * 64 + (((lane >> 4) + 2 * ((lane & 15) >> 2)) & 3) * 16) % 256 // 16 for lane in grp}
        assert len(slots) == 16


def test_fused_cross_attention_image_index_maps():
    """Round 6 (attention.py:163-194, 273): `ops.pack_xattn_wq / _wo / _kv` against the data flow of csrc/xattn_fused.hip restated with index arithmetic on the CPU, for one
    16-row fragment.  MFMA facts used (the ones csrc/ff_fused.hip rests on): D[i][n] = sum over (g, e) of A(lane (i, g), e) B(lane (n, g), e); a result fragment leaves lane
    (n, g') with rows i = 4 g' + r.  Chain: Q (Wq image, XOR-swizzled rows) -> q slots 16 nf + 4 g + r -> logits against the K image (K step t, element e <-> slot
    16 (2 t + (e >> 2)) + 4 g + (e & 3)) -> probabilities in the same element order against the V^T image (row 40 = ones) -> two heads' outputs against the Wo image."""
    from anyedit_amd import ops
    gen = torch.Generator().manual_seed(9)
    B, Nk, T, R, C, H, D = 2, 78, 4, 16, 320, 8, 40
    bf = lambda t_: t_.to(torch.bfloat16).float()  # noqa: E731
    wq, wo = bf(torch.randn(C, C, generator=gen) / 18), bf(torch.randn(C, C, generator=gen) / 18)
    kv, kv_ip = bf(torch.randn(B * Nk, 2 * C, generator=gen)), bf(torch.randn(B * T, 2 * C, generator=gen))
    xn = bf(torch.randn(R, C, generator=gen))
    gate = 0.7
    wq_img, wo_img = ops.pack_xattn_wq(wq).float(), ops.pack_xattn_wo(wo).float()
    kv_img = ops.pack_xattn_kv(kv.to(torch.bfloat16), kv_ip.to(torch.bfloat16), B, Nk, T).float()
    b = 1
    scale = D ** -0.5
    out = torch.zeros(R, C)
    o_pair = {}
    for h in range(H):
        # ---- Q: lane (i = l15, g) of fragment nf reads piece 4 ks + g of image row 16 nf + l15 at position (c & ~7) | ((c ^ (l15 >> 1)) & 7)
        qslot = torch.zeros(48, R)
        for nf in range(3):
            for i in range(16):
                row = wq_img[h, 16 * nf + i].reshape(40, 8)
                w = torch.stack([row[(c & ~7) | ((c ^ (i >> 1)) & 7)] for c in range(40)]).reshape(320)   # the logical row the reads reassemble
                qslot[16 * nf + i] = xn @ w
        assert float((qslot[:40] - (xn @ wq[40 * h:40 * h + 40].t()).t()).abs().max()) < 1e-4 and float(qslot[40:].abs().max()) == 0.0
        kimg = kv_img[b, h, :7680].reshape(96, 80)
        vimg = kv_img[b, h, 7680:7680 + 6912].reshape(48, 144)
        # ---- logits: B operand element (t, g, e) = q slot 16 (2 t + (e >> 2)) + 4 g + (e & 3) (zero from slot 48 on)
        S = torch.zeros(96, R)
        for t in range(2):
            for g in range(4):
                for e in range(8):
                    slot = 16 * (2 * t + (e >> 2)) + 4 * g + (e & 3)
                    if slot < 48:
                        S += kimg[:, 32 * t + 8 * g + e][:, None] * qslot[slot][None, :]
        S = S * scale
        kb = kv[b * Nk:(b + 1) * Nk].reshape(Nk, 2, H, D)
        kib = kv_ip[b * T:(b + 1) * T].reshape(T, 2, H, D)
        ref_s = (xn @ wq[40 * h:40 * h + 40].t()) @ kb[:, 0, h].t() * scale
        assert float((S[:Nk].t() - ref_s).abs().max()) < 1e-3 and float(S[Nk:80].abs().max()) == 0.0
        P1 = torch.zeros(96, R)
        P1[:Nk] = torch.softmax(S[:Nk], 0)
        P2 = torch.zeros(16, R)
        P2[:T] = torch.softmax(S[80:80 + T], 0)
        # ---- PV: K steps 0 .. 2 carry text keys 16 (2 t + j) + 4 g + r (j = 1 of step 2: zeros), step 3 expert keys 4 g + r (j = 0)
        O1, O2 = torch.zeros(48, R), torch.zeros(48, R)
        for t in range(4):
            for g in range(4):
                for e in range(8):
                    j, r = e >> 2, e & 3
                    a = vimg[:, 32 * t + 8 * g + e]
                    if t < 3:
                        key = 16 * (2 * t + j) + 4 * g + r
                        pk = P1[key] if (key < 80 and not (t == 2 and j == 1)) else torch.zeros(R)
                        O1 += a[:, None] * pk[None, :]
                    elif j == 0:
                        O2 += a[:, None] * P2[4 * g + r][None, :]
        o = O1 / O1[40] + gate * O2 / O2[40]                    # row 40: the denominators
        ref_o = torch.softmax(ref_s, -1) @ kb[:, 1, h] + gate * torch.softmax((xn @ wq[40 * h:40 * h + 40].t()) @ kib[:, 0, h].t() * scale, -1) @ kib[:, 1, h]
        assert float((o[:40].t() - ref_o).abs().max()) < 1e-3
        o_pair[h & 1] = o
        if h & 1:
            # ---- output projection: K step t, element e <-> fragment q6 = 2 t + (e >> 2) of the pair: head q6 >= 3, d slot 16 (q6 % 3) + 4 g + (e & 3)
            img = wo_img[h >> 1]
            D_ = torch.zeros(320, R)
            for t in range(3):
                for g in range(4):
                    for e in range(8):
                        q6 = 2 * t + (e >> 2)
                        D_ += img[:, 32 * t + 8 * g + e][:, None] * o_pair[int(q6 >= 3)][16 * (q6 % 3) + 4 * g + (e & 3)][None, :]
            for cf in range(20):
                for gp in range(4):
                    for r in range(4):
                        out[:, 32 * (cf >> 1) + 8 * gp + 4 * (cf & 1) + r] += D_[16 * cf + 4 * gp + r]
    qf = (xn @ wq.t()).reshape(R, H, D)
    kb = kv[b * Nk:(b + 1) * Nk].reshape(Nk, 2, H, D)
    kib = kv_ip[b * T:(b + 1) * T].reshape(T, 2, H, D)
    att = torch.einsum("rhk,khd->rhd", torch.softmax(torch.einsum("rhd,khd->rhk", qf, kb[:, 0]) * scale, -1), kb[:, 1]) + \
        gate * torch.einsum("rhk,khd->rhd", torch.softmax(torch.einsum("rhd,khd->rhk", qf, kib[:, 0]) * scale, -1), kib[:, 1])
    ref = att.reshape(R, C) @ wo.t()
    assert float((out - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
    # conflict-free fragment reads: row strides 160 / 288 / 224 bytes put the 16 lanes of a ds_read_b128 group on 16 different 16-byte slots of the 256-byte bank row
    for stride in (160, 288, 224):
        for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
            assert len({((lane & 15) * stride + (lane >> 4) * 16) % 256 // 16 for lane in grp}) == 16, stride


def test_mask_tool_box_logic_on_cpu():
    """tools/tool.py:184-222 (no GPU involved): box conversion and the phrase-based target filter, including its fallbacks and the
    list form of `target_object`."""
    from anyedit_amd.tools.tool import boxes_to_pixels_xyxy, select_target_boxes, load_image_512
    from PIL import Image
    dets = torch.tensor([[0.30, 0.40, 0.30, 0.40], [0.75, 0.70, 0.20, 0.30], [0.5, 0.5, 0.9, 0.9]])
    phrases = ["cat(0.81)", "black cat(0.55)", "red sofa(0.90)"]
    px = boxes_to_pixels_xyxy(dets, 512, 256)
    assert torch.allclose(px[0], torch.tensor([0.15 * 512, 0.2 * 256, 0.45 * 512, 0.6 * 256]))
    assert torch.equal(dets, torch.tensor([[0.30, 0.40, 0.30, 0.40], [0.75, 0.70, 0.20, 0.30], [0.5, 0.5, 0.9, 0.9]]))   # input not mutated
    b, s = select_target_boxes(px, phrases, "cat")
    assert b.shape == (1, 4) and s.tolist() == pytest.approx([0.81])                      # exact name match wins
    b, s = select_target_boxes(px, phrases, "black cat")
    assert b.shape == (1, 4) and s.tolist() == pytest.approx([0.55])
    b, s = select_target_boxes(px, phrases, "fluffy cat")                                  # no exact match -> word overlap: both cats
    assert b.shape == (2, 4)
    b, s = select_target_boxes(px, phrases, ["dog", "red sofa"])
    assert b.shape == (1, 4) and s.tolist() == pytest.approx([0.90])
    assert select_target_boxes(px, phrases, ["dog", "lamp"]) is None
    img = load_image_512(Image.new("L", (300, 200), 128))
    assert img.size == (512, 512) and img.mode == "RGB"


def test_packed_weight_caches_follow_the_parameters(tmp_path):
    """packed bf16 weight caches must not survive a load_state_dict on a PARENT module, an in-place parameter update or a
    device move.  The caches are keyed on (storage address, version counter) of the tensors they were built from."""
    import torch.nn as nn
    from anyedit_amd.ldm.modules.diffusionmodules.util import Linear, GroupNorm32
    from anyedit_amd.ldm.modules.attention import CrossAttention

    class Parent(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin, self.attn, self.gn = Linear(16, 8), CrossAttention(16, heads=2, dim_head=8), GroupNorm32(4, 16)

    torch.manual_seed(0)
    p = Parent()
    w0, q0 = p.lin._packed()["w"].clone(), p.attn._packed()["q"].clone()
    assert p.lin._packed() is p.lin._packed()                      # a second call serves the cache
    sd = {k: torch.randn_like(v) for k, v in p.state_dict().items()}
    p.load_state_dict(sd)                                           # the parent's load never calls a child's load_state_dict override
    assert torch.equal(p.lin._packed()["w"].float(), sd["lin.weight"].to(torch.bfloat16).float()) and not torch.equal(p.lin._packed()["w"], w0)
    assert torch.equal(p.attn._packed()["q"].float(), sd["attn.to_q.weight"].to(torch.bfloat16).float()) and not torch.equal(p.attn._packed()["q"], q0)
    assert torch.equal(p.gn._affine()[0], sd["gn.weight"])
    with torch.no_grad():
        p.lin.weight.mul_(2.0)                                      # optimiser-style in-place update
    assert torch.equal(p.lin._packed()["w"].float(), (sd["lin.weight"] * 2.0).to(torch.bfloat16).float())


def test_checkpoint_loader_weights_only_then_trusted_fallback(tmp_path, monkeypatch):
    """torch >= 2.6 unpickles with weights_only=True; a Lightning-style .ckpt with non-tensor objects needs an explicit opt-in."""
    import argparse
    from anyedit_amd.cldm.model import load_state_dict, trusted_torch_load
    plain = tmp_path / "plain.ckpt"
    torch.save({"state_dict": {"a.weight": torch.ones(2, 2)}}, plain)
    assert torch.equal(load_state_dict(str(plain))["a.weight"], torch.ones(2, 2))
    lightning = tmp_path / "lightning.ckpt"
    torch.save({"state_dict": {"a.weight": torch.ones(2, 2)}, "hyper_parameters": argparse.Namespace(lr=1e-4), "callbacks": {object: 1}}, lightning)
    monkeypatch.delenv("ANYEDIT_TRUST_CHECKPOINTS", raising=False)
    with pytest.raises(RuntimeError, match="ANYEDIT_TRUST_CHECKPOINTS"):
        trusted_torch_load(lightning)
    monkeypatch.setenv("ANYEDIT_TRUST_CHECKPOINTS", "1")
    assert torch.equal(load_state_dict(str(lightning))["a.weight"], torch.ones(2, 2))


def test_mask_blend_accepts_every_mask_that_broadcasts():
    """the reference's `img_orig * mask + (1 - mask) * img` (ddim.py:154-157) takes any broadcastable mask; the kernel reads
    one plane per sample, so a per-channel mask runs as B*C single-channel samples."""
    from anyedit_amd import ops
    B, C, H, W = 2, 4, 3, 5
    m, Bk, Ck = ops.mask_blend_shape(torch.rand(B, 1, H, W), B, C, H, W)
    assert (tuple(m.shape), Bk, Ck) == ((B, 1, H, W), B, C)
    m, Bk, Ck = ops.mask_blend_shape(torch.rand(1, 1, H, W), B, C, H, W)
    assert (tuple(m.shape), Bk, Ck) == ((B, 1, H, W), B, C)
    mc = torch.rand(B, C, H, W)
    m, Bk, Ck = ops.mask_blend_shape(mc, B, C, H, W)
    assert (tuple(m.shape), Bk, Ck) == ((B * C, 1, H, W), B * C, 1) and torch.equal(m.reshape(B, C, H, W), mc)
    m, Bk, Ck = ops.mask_blend_shape(torch.rand(H, W), B, C, H, W)
    assert (tuple(m.shape), Bk, Ck) == ((B, 1, H, W), B, C)
    with pytest.raises(ValueError):
        ops.mask_blend_shape(torch.rand(B, 3, H, W), B, C, H, W)


def test_c_abi_rejects_bad_arguments_before_touching_the_gpu():
    """Error behaviour of the drop-in boundary: every entry point validates its arguments on the host first and returns AE_ERR_ARG (-1)
    with a message behind ae_last_error() — no launch is attempted, so this runs without a GPU.  (Empty inputs, misaligned rows,
    unsupported sizes: the cases the reference would surface as Python exceptions from torch.)"""
    import ctypes
    from anyedit_amd._lib import lib
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) // 16 * 16

    def expect(rc, fragment):
        assert rc == -1, rc
        msg = lib.ae_last_error().decode()
        assert fragment in msg, msg

    expect(lib.ae_gemm_bf16(None, 64, None, 0, 0, p16, 64, p16, 64, 8, 8, 64, None, None, 0, None, 0, 0, 0, 0, None, None), "null pointer")
    expect(lib.ae_gemm_bf16(p16, 64, None, 0, 0, p16, 64, p16, 64, 0, 8, 64, None, None, 0, None, 0, 0, 0, 0, None, None), "must be positive")       # empty M
    expect(lib.ae_gemm_bf16(p16, 64, None, 0, 0, p16, 64, p16, 64, 8, 8, 60, None, None, 0, None, 0, 0, 0, 0, None, None), "multiple of 8")           # ragged K
    expect(lib.ae_gemm_bf16(p16 + 2, 64, None, 0, 0, p16, 64, p16, 64, 8, 8, 64, None, None, 0, None, 0, 0, 0, 0, None, None), "16-byte aligned")    # misaligned A
    expect(lib.ae_gemm_bf16(p16, 64, None, 0, 0, p16, 64, p16, 64, 8, 8, 64, None, None, 0, None, 0, 0, 9, 0, None, None), "bad epilogue")
    expect(lib.ae_conv3x3_bf16(p16, p16, None, None, 0, None, p16, 0, 8, 8, 64, 64, 1, 0, 0, None, None, 0, None), "bad shape")                           # empty batch
    expect(lib.ae_conv3x3_bf16(p16, p16, None, None, 0, None, p16, 1, 8, 8, 60, 64, 1, 0, 0, None, None, 0, None), "multiple of 8")
    expect(lib.ae_conv3x3_bf16(p16, p16, None, None, 0, None, p16, 1, 8, 8, 64, 64, 3, 0, 0, None, None, 0, None), "stride must be 1 or 2")
    expect(lib.ae_groupnorm_nhwc_bf16(p16, None, 0, p16, p16, p16, 1, 64, 60, 32, 1e-5, 0, p16, None, None, None, None, None), "bad shape")                  # C % groups
    expect(lib.ae_groupnorm_nhwc_bf16(p16, None, 0, p16, p16, p16, 1, 64, 64, 32, 1e-5, 7, p16, None, None, None, None, None), "act must be")
    expect(lib.ae_layernorm_bf16(p16, p16, p16, p16, 4, 60, 1e-5, None), "multiple of 8")
    expect(lib.ae_expert_kv_fwd(p16, p16, p16, p16, 2, 9, 64, 64, 2, None), "unsupported shape")                                                 # more than 8 tokens per sample
    expect(lib.ae_split_channels_bf16(p16, 12, 8, p16, p16, 4, 0, 0, None), "bad arguments")
    assert lib.ae_ln_gemm_supported(49152, 320, 320, 0) == 1 and lib.ae_ln_gemm_supported(49152, 320, 640, 0) == 0


def test_diffusion_wrapper_routes_every_conditioning_key():
    """DiffusionWrapper.forward / LatentDiffusion.apply_model (ddpm.py:1336-1363, 854-869): which tensors reach the UNet for each
    conditioning key — channel concat, context concat or list (sequential_crossattn), class vector from c_adm or c_crossattn[0]."""
    from anyedit_amd.ldm.models.diffusion.ddpm import DiffusionWrapper, LatentDiffusion

    class Probe(torch.nn.Module):
        def forward(self, x, t, context=None, y=None):
            self.seen = (x, context, y)
            return x[:, :1]

    x, t = torch.zeros(2, 4, 3, 3), torch.zeros(2, dtype=torch.long)
    cc = [torch.ones(2, 4, 3, 3), 2 * torch.ones(2, 1, 3, 3)]
    ca = [torch.ones(2, 5, 8), 2 * torch.ones(2, 3, 8)]
    adm = torch.arange(2.)
    for key, want_c, want_ctx, want_y in ((None, 4, None, None), ("concat", 9, None, None), ("crossattn", 4, 8, None),
                                          ("hybrid", 9, 8, None), ("hybrid-adm", 9, 8, adm), ("crossattn-adm", 4, 8, adm),
                                          ("adm", 4, None, ca[0])):
        w = DiffusionWrapper(Probe(), key)
        out = w(x, t, c_concat=cc, c_crossattn=ca, c_adm=adm)
        sx, sctx, sy = w.diffusion_model.seen
        assert out.shape == (2, 1, 3, 3) and sx.shape[1] == want_c, key
        if want_c == 9:
            assert torch.equal(sx[:, 4:8], cc[0]) and torch.equal(sx[:, 8:], cc[1])
        assert (sctx is None) == (want_ctx is None) and (sctx is None or (sctx.shape == (2, want_ctx, 8) and torch.equal(sctx[:, 5:], ca[1]))), key
        assert (sy is None) == (want_y is None) and (sy is None or sy is want_y), key
    for key in ("hybrid-adm", "crossattn-adm"):
        with pytest.raises(AssertionError):
            DiffusionWrapper(Probe(), key)(x, t, c_concat=cc, c_crossattn=ca)
    with pytest.raises(AssertionError):
        DiffusionWrapper(Probe(), "bogus")
    w = DiffusionWrapper(Probe(), "crossattn")
    w.sequential_cross_attn = True
    w(x, t, c_crossattn=ca)
    assert w.diffusion_model.seen[1] is ca                      # the list itself, not a concatenation
    # apply_model: a bare tensor or list lands in the slot the key reads; tuples are unwrapped unless return_ids
    ldm = LatentDiffusion(Probe(), conditioning_key="crossattn")
    ldm.apply_model(x, t, ca[0])
    assert torch.equal(ldm.model.diffusion_model.seen[1], ca[0])
    ldm.apply_model(x, t, ca)
    assert ldm.model.diffusion_model.seen[1].shape == (2, 8, 8)
    ldm = LatentDiffusion(Probe(), conditioning_key="concat")
    ldm.apply_model(x, t, cc[0])
    assert ldm.model.diffusion_model.seen[0].shape[1] == 8
    ldm.model.forward = lambda *a, **k: ("eps", "ids")
    assert ldm.apply_model(x, t, {}) == "eps" and ldm.apply_model(x, t, {}, return_ids=True) == ("eps", "ids")
    # first-stage glue: frozen, eval, scale factor on encode / its inverse on decode
    class VAE(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.ones(1))
        def decode(self, z):
            return z
        def encode(self, x):
            return x
    ldm = LatentDiffusion(Probe(), scale_factor=0.5, first_stage_config=VAE())
    assert not ldm.first_stage_model.training and not ldm.first_stage_model.p.requires_grad
    assert float(ldm.decode_first_stage(torch.ones(1))) == 2.0 and float(ldm.get_first_stage_encoding(torch.ones(1))) == 0.5
    with pytest.raises(NotImplementedError):
        ldm.get_first_stage_encoding("latent")
