"""GPU parity tests for SURVEY.md §8(f) N3: SAM prompt encoder, two-way transformer mask decoder, mask post-processing and the
SamPredictor front end, against the reference-derived golden (tests/golden/sam_decoder.npz) and the oracle restatement."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import load_golden, sub_sd, T, rel_l2, psnr

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def close(got, ref, rl2=2e-2, db=36.0, what=""):
    got, ref = got.detach().float().cpu(), torch.as_tensor(ref).float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    e, p = rel_l2(got, ref), psnr(got, ref)
    assert e <= rl2 and p >= db, f"{what}: rel_l2={e:.3e} (<= {rl2}), psnr={p:.1f} dB (>= {db})"


def close_abs(got, ref, atol=5e-3, rtol=3e-2, what=""):
    """For the few-element IoU predictions (PSNR against a 2-value dynamic range says nothing)."""
    got, ref = got.detach().float().cpu(), torch.as_tensor(ref).float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert bool(((got - ref).abs() <= atol + rtol * ref.abs()).all()), f"{what}: {got.flatten().tolist()} vs {ref.flatten().tolist()}"


def tiny_sam():
    from anyedit_amd.segment_anything.modeling import ImageEncoderViT, MaskDecoder, PromptEncoder, Sam, TwoWayTransformer
    enc = ImageEncoderViT(img_size=128, patch_size=16, in_chans=3, embed_dim=64, depth=2, num_heads=2, mlp_ratio=4.0, out_chans=64,
                          qkv_bias=True, norm_layer=lambda d: nn.LayerNorm(d, eps=1e-6), use_abs_pos=True, use_rel_pos=True,
                          window_size=4, global_attn_indexes=(1,))
    pe = PromptEncoder(embed_dim=64, image_embedding_size=(8, 8), input_image_size=(128, 128), mask_in_chans=16)
    md = MaskDecoder(num_multimask_outputs=3, transformer=TwoWayTransformer(depth=2, embedding_dim=64, mlp_dim=128, num_heads=4),
                     transformer_dim=64, iou_head_depth=3, iou_head_hidden_dim=64)
    return Sam(image_encoder=enc, prompt_encoder=pe, mask_decoder=md).eval()


def golden_sam():
    g = load_golden("sam_decoder")
    sam = tiny_sam()
    missing, unexpected = sam.load_state_dict(sub_sd(g, "w."), strict=False)
    assert not unexpected and all(k.startswith("image_encoder.") for k in missing), (missing, unexpected)
    return g, sam.to(DEV)


def cases(g):
    pts, lbl = T(g["point_coords"]).to(DEV), T(g["point_labels"]).to(DEV)
    return {"boxes": (None, T(g["boxes"]).to(DEV), None, False), "points": ((pts, lbl), None, None, True),
            "all": ((pts, lbl), T(g["boxes"])[:2].to(DEV), T(g["mask_input"]).to(DEV), True)}


# ------------------------------------------------------------------------------------------------------------ kernels
def test_pe_encode_and_prompt_embeddings_golden():
    g, sam = golden_sam()
    pe = sam.prompt_encoder
    close(pe.get_dense_pe(), g["dense_pe"], rl2=1e-5, db=90.0, what="dense_pe")
    for tag, (points, boxes, masks, _) in cases(g).items():
        sparse, dense = pe(points, boxes, masks)
        close(sparse, g[f"{tag}.sparse"], rl2=1e-5, db=90.0, what=f"{tag}.sparse")
        if masks is not None:
            close(dense, g[f"{tag}.dense"], rl2=1e-2, db=40.0, what=f"{tag}.dense")
        else:
            assert dense.shape == (sparse.shape[0], 64, 8, 8) and dense.stride(0) == 0
    with pytest.raises(ValueError):
        pe._embed_points(torch.zeros(1, 1, 2, device=DEV), torch.full((1, 1), 2, device=DEV), pad=True)


def test_pe_encode_large_coordinates_vs_oracle():
    """Full-size geometry: 1024-px frame, 128 frequencies, coordinates across and beyond the frame."""
    from anyedit_amd import ops
    from oracle import sam_decoder_ref as SD
    gen = torch.Generator().manual_seed(5)
    gauss = torch.randn(2, 128, generator=gen)
    coords = torch.rand(1000, 2, generator=gen) * 1100.0 - 30.0
    ref = SD.pe_with_coords({"prompt_encoder.pe_layer.positional_encoding_gaussian_matrix": gauss}, coords[None] + 0.5, (1024, 1024))[0]
    out = ops.sam_pe_encode(coords.to(DEV), gauss.to(DEV), (1024, 1024), offset=0.5)
    assert float((out.cpu() - ref).abs().max()) < 2e-5


def test_mask_downscale_kernel_vs_oracle():
    from anyedit_amd import ops
    from oracle import sam_decoder_ref as SD
    g, sam = golden_sam()
    sd = sub_sd(g, "w.")
    gen = torch.Generator().manual_seed(6)
    masks = torch.randn(3, 1, 64, 96, generator=gen) * 5.0           # non-square 16 x 24 embedding grid
    q = "prompt_encoder.mask_downscaling."
    h = F.conv2d(masks, sd[q + "0.weight"], sd[q + "0.bias"], stride=2)
    h = F.gelu(SD.layer_norm_2d(h, sd[q + "1.weight"], sd[q + "1.bias"]))
    h = F.conv2d(h, sd[q + "3.weight"], sd[q + "3.bias"], stride=2)
    ref = F.gelu(SD.layer_norm_2d(h, sd[q + "4.weight"], sd[q + "4.bias"]))
    pk = sam.prompt_encoder._packed()["md"]
    rows = ops.sam_mask_downscale(masks.to(DEV), *pk, eps=1e-6)
    got = rows.float().cpu().reshape(3, 16, 24, 16).permute(0, 3, 1, 2)
    close(got, ref, rl2=4e-3, db=48.0, what="mask_downscale")


@pytest.mark.parametrize("C", [8, 16, 64, 104, 256, 512])
def test_layernorm_act_vs_torch(C):
    from anyedit_amd import ops
    gen = torch.Generator().manual_seed(C)
    for M in (1, 37, 4099):
        x = (torch.randn(M, C, generator=gen) * 2 + 0.5).to(BF)
        w, b = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
        for gelu in (True, False):
            ref = F.layer_norm(x.float(), (C,), w, b, 1e-6)
            ref = F.gelu(ref) if gelu else ref
            out = ops.layernorm_act(x.to(DEV), w.to(DEV), b.to(DEV), eps=1e-6, gelu=gelu)
            close(out, ref, rl2=4e-3, db=46.0, what=f"layernorm_act C={C} M={M} gelu={gelu}")


@pytest.mark.parametrize("C,M", [(8, 4), (32, 4), (32, 1), (64, 3)])
def test_mask_product_unshuffle_vs_torch(C, M):
    """up is the un-shuffled output of two k2s2 transposed convolutions: [b, y, x, dy1, dx1, dy2, dx2, c]."""
    from anyedit_amd import ops
    gen = torch.Generator().manual_seed(C + M)
    B, h, w = 2, 5, 7
    up = torch.randn(B, h, w, 2, 2, 2, 2, C, generator=gen).to(BF)
    hyper = torch.randn(B, M, C, generator=gen)
    img = up.float().permute(0, 7, 1, 3, 5, 2, 4, 6).reshape(B, C, 4 * h, 4 * w)      # (b, c, y, dy1, dy2, x, dx1, dx2)
    ref = (hyper @ img.reshape(B, C, -1)).reshape(B, M, 4 * h, 4 * w)
    out = ops.sam_mask_product(up.reshape(-1, C).to(DEV), hyper.to(DEV), B, h, w)
    assert float((out.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("geom", [((32, 32), 128, (96, 128), (75, 100)), ((256, 256), 1024, (683, 1024), (512, 768)),
                                  ((256, 256), 1024, (1024, 1024), (1536, 1536)), ((256, 256), 1024, (1024, 768), (333, 250))])
def test_postprocess_masks_vs_oracle(geom):
    from anyedit_amd import ops
    from oracle import sam_decoder_ref as SD
    (Hl, Wl), S, inp, orig = geom
    gen = torch.Generator().manual_seed(Hl + orig[0])
    low = torch.randn(2, 3, Hl, Wl, generator=gen) * 8.0
    ref = SD.postprocess_masks(low, S, inp, orig)
    logits, binary = ops.sam_postprocess_masks(low.to(DEV), S, inp, orig, threshold=0.0, want_logits=True)
    assert float((logits.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    disagree = (binary.cpu() != (ref > 0.0)) & (ref.abs() > 1e-4)          # only exact-zero crossings may differ
    assert binary.dtype == torch.bool and int(disagree.sum()) == 0
    only_mask = ops.sam_postprocess_masks(low.to(DEV), S, inp, orig, threshold=0.0, want_logits=False)
    assert only_mask[0] is None and torch.equal(only_mask[1], binary)


def test_preprocess_golden_and_uint8():
    g, sam = golden_sam()
    x = T(g["pre.x"]).to(DEV)
    y = sam.preprocess(x)
    assert float((y.cpu() - T(g["pre.y"])).abs().max()) < 1e-5
    assert torch.equal(sam.preprocess(x.to(torch.uint8)), y)
    assert sam.preprocess(x[0]).shape == (3, 128, 128)


# ------------------------------------------------------------------------------------------------------------ modules
def test_two_way_transformer_and_decoder_golden():
    """Prompt encoder -> MaskDecoder.forward (public NCHW API) -> postprocess_masks for the three prompt mixes of the golden."""
    g, sam = golden_sam()
    emb = T(g["image_embedding"]).to(DEV)
    for tag, (points, boxes, masks, multi) in cases(g).items():
        sparse, dense = sam.prompt_encoder(points, boxes, masks)
        low, iou = sam.mask_decoder(image_embeddings=emb, image_pe=sam.prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=sparse,
                                    dense_prompt_embeddings=dense, multimask_output=multi)
        close(low, g[f"{tag}.low_res"], rl2=3e-2, db=34.0, what=f"{tag}.low_res")
        close_abs(iou, g[f"{tag}.iou"], what=f"{tag}.iou")
        full = sam.postprocess_masks(low, (96, 128), (75, 100))
        close(full, g[f"{tag}.masks"], rl2=3e-2, db=34.0, what=f"{tag}.masks")
        ref_bin = T(g[f"{tag}.masks"]) > 0
        agree = float(((full.cpu() > 0) == ref_bin).float().mean())
        assert agree > 0.985, (tag, agree)


def test_decoder_public_modules_vs_oracle():
    """The nn.Module forwards of the decoder building blocks (reference call signatures) against the oracle."""
    from oracle import sam_decoder_ref as SD
    g, sam = golden_sam()
    sd = sub_sd(g, "w.")
    gen = torch.Generator().manual_seed(9)
    tr = sam.mask_decoder.transformer
    q, k, v = torch.randn(2, 7, 64, generator=gen), torch.randn(2, 64, 64, generator=gen), torch.randn(2, 64, 64, generator=gen)
    p = "mask_decoder.transformer.layers.1.cross_attn_token_to_image."
    close(tr.layers[1].cross_attn_token_to_image(q.to(DEV), k.to(DEV), v.to(DEV)), SD.attention(sd, p, q, k, v, 4), what="Attention t2i")
    p = "mask_decoder.transformer.layers.0.self_attn."
    close(tr.layers[0].self_attn(q.to(DEV), q.to(DEV), q.to(DEV)), SD.attention(sd, p, q, q, q, 4), what="Attention self")
    pe = torch.randn(1, 64, 64, generator=gen).expand(2, -1, -1)
    qpe = torch.randn(2, 7, 64, generator=gen)
    for i in (0, 1):
        rq, rk = SD.two_way_block(sd, f"mask_decoder.transformer.layers.{i}.", q, k, qpe, pe, 4, i == 0)
        oq, ok = tr.layers[i](q.to(DEV), k.to(DEV), qpe.to(DEV), pe.to(DEV))
        close(oq, rq, what=f"block{i}.queries")
        close(ok, rk, what=f"block{i}.keys")
    img, ipe = torch.randn(2, 64, 8, 8, generator=gen), torch.randn(1, 64, 8, 8, generator=gen)
    rq, rk = SD.two_way_transformer(sd, "mask_decoder.transformer.", img, ipe.expand(2, -1, -1, -1), qpe, 2, 4)
    oq, ok = tr(img.to(DEV), ipe.to(DEV), qpe.to(DEV))
    close(oq, rq, what="transformer.queries")
    close(ok, rk, what="transformer.keys")
    x = torch.randn(5, 64, generator=gen)
    close(sam.mask_decoder.iou_prediction_head(x.to(DEV)), SD.mlp(sd, "mask_decoder.iou_prediction_head.", x), what="MLP")
    up0 = sam.mask_decoder.output_upscaling[0]
    xin = torch.randn(2, 64, 5, 6, generator=gen)
    ref = F.conv_transpose2d(xin, sd["mask_decoder.output_upscaling.0.weight"], sd["mask_decoder.output_upscaling.0.bias"], stride=2)
    close(up0(xin.to(DEV)), ref, what="ConvTranspose2x2")


def test_sam_predictor_end_to_end_vs_oracle():
    """SamPredictor.set_image -> predict_torch (boxes, as tools/tool.py:227-237) and predict (points) on a seeded tiny SAM, against the
    oracle chain image_encoder -> prompt_encoder -> mask_decoder -> postprocess_masks run on the same resized image."""
    from anyedit_amd.segment_anything import SamPredictor
    from oracle import sam_ref as M, sam_decoder_ref as SD
    torch.manual_seed(11)
    sam = tiny_sam()
    gen = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for p in sam.parameters():
            if p.abs().sum() == 0:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
    sd = {k: v.detach().clone().float() for k, v in sam.state_dict().items()}
    sam = sam.to(DEV)
    pred = SamPredictor(sam)
    with pytest.raises(RuntimeError):
        pred.predict_torch(None, None, boxes=torch.zeros(1, 4, device=DEV))
    image = torch.randint(0, 256, (75, 100, 3), generator=gen, dtype=torch.uint8).numpy()
    pred.set_image(image)
    assert pred.input_size == (96, 128) and tuple(pred.original_size) == (75, 100)
    assert pred.get_image_embedding().shape == (1, 64, 8, 8)

    resized = torch.from_numpy(pred.transform.apply_image(image)).permute(2, 0, 1)[None].float()
    mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
    emb = M.image_encoder(sd, "image_encoder.", SD.preprocess(resized, 128, mean, std), 16, 2, 2, 4, (1,))
    close(pred.get_image_embedding(), emb, rl2=3e-2, db=32.0, what="embedding")
    emb = pred.get_image_embedding().float().cpu()     # decoder parity is measured from the same embedding

    boxes = torch.tensor([[10.0, 8.0, 60.0, 70.0], [30.0, 20.0, 99.0, 74.0]])
    tb = pred.transform.apply_boxes_torch(boxes, image.shape[:2])
    masks, iou, low = pred.predict_torch(None, None, boxes=tb.to(DEV), multimask_output=False)
    sparse, dense = SD.prompt_encoder(sd, None, tb, None, (8, 8), (128, 128))
    rlow, riou = SD.mask_decoder(sd, emb, SD.dense_pe(sd, (8, 8)), sparse, dense, False, 2, 4)
    rfull = SD.postprocess_masks(rlow, 128, (96, 128), (75, 100))
    assert masks.dtype == torch.bool and masks.shape == (2, 1, 75, 100) and low.shape == (2, 1, 32, 32)
    close(low, rlow, rl2=3e-2, db=34.0, what="predict_torch.low_res")
    close_abs(iou, riou, what="predict_torch.iou")
    assert float((masks.cpu() == (rfull > 0)).float().mean()) > 0.985
    logits, _, _ = pred.predict_torch(None, None, boxes=tb.to(DEV), multimask_output=False, return_logits=True)
    close(logits, rfull, rl2=3e-2, db=34.0, what="predict_torch.logits")

    pts, lbl = np.array([[50.0, 40.0], [10.0, 60.0]]), np.array([1, 0])
    m, i, l = pred.predict(point_coords=pts, point_labels=lbl, multimask_output=True, mask_input=low[0].float().cpu().numpy())
    tp = torch.as_tensor(pred.transform.apply_coords(pts, image.shape[:2]), dtype=torch.float)[None]
    sparse, dense = SD.prompt_encoder(sd, (tp, torch.as_tensor(lbl)[None]), None, low[:1].float().cpu(), (8, 8), (128, 128))
    rlow, riou = SD.mask_decoder(sd, emb, SD.dense_pe(sd, (8, 8)), sparse, dense, True, 2, 4)
    assert m.shape == (3, 75, 100) and m.dtype == np.bool_ and i.shape == (3,) and l.shape == (3, 32, 32)
    close(torch.from_numpy(l), rlow[0], rl2=3e-2, db=34.0, what="predict.low_res")

    out = sam([{"image": resized[0].to(DEV), "original_size": (75, 100), "boxes": tb.to(DEV)}], multimask_output=False)[0]
    assert float((out["masks"] == masks).float().mean()) > 0.999 and out["iou_predictions"].shape == (2, 1)


def test_sam_decoder_full_size_geometry_runs():
    """build_sam's decoder at its real sizes (256-wide, 64x64 embedding, 8 heads of 32 / 16, 5 boxes) — finite, right shapes, and the
    low-res masks agree with the oracle from the same embedding."""
    from anyedit_amd.segment_anything.modeling import MaskDecoder, PromptEncoder, TwoWayTransformer
    from oracle import sam_decoder_ref as SD
    torch.manual_seed(21)
    pe = PromptEncoder(embed_dim=256, image_embedding_size=(64, 64), input_image_size=(1024, 1024), mask_in_chans=16)
    md = MaskDecoder(num_multimask_outputs=3, transformer=TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
                     transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256)
    sd = {"prompt_encoder." + k: v.detach().clone().float() for k, v in pe.state_dict().items()}
    sd.update({"mask_decoder." + k: v.detach().clone().float() for k, v in md.state_dict().items()})
    pe, md = pe.to(DEV), md.to(DEV)
    gen = torch.Generator().manual_seed(22)
    emb = torch.randn(1, 256, 64, 64, generator=gen) * 0.5
    boxes = torch.rand(5, 2, 2, generator=gen).sort(dim=1).values.reshape(5, 4) * 1000.0
    sparse, dense = pe(None, boxes.to(DEV), None)
    low, iou = md(image_embeddings=emb.to(DEV), image_pe=pe.get_dense_pe(), sparse_prompt_embeddings=sparse,
                  dense_prompt_embeddings=dense, multimask_output=False)
    rs, rd = SD.prompt_encoder(sd, None, boxes, None, (64, 64), (1024, 1024))
    rlow, riou = SD.mask_decoder(sd, emb, SD.dense_pe(sd, (64, 64)), rs, rd, False, 2, 8)
    assert low.shape == (5, 1, 256, 256) and iou.shape == (5, 1)
    close(low, rlow, rl2=4e-2, db=32.0, what="full-size low_res")
    close_abs(iou, riou, what="full-size iou")


# ------------------------------------------------------------------------------------------------------------ mask tool
@pytest.mark.parametrize("N", [1, 7, 300, 1100])
def test_nms_vs_restatement(N):
    from anyedit_amd import ops
    from oracle import sam_decoder_ref as SD
    gen = torch.Generator().manual_seed(N)
    c = torch.rand(N, 2, generator=gen) * 400
    wh = torch.rand(N, 2, generator=gen) * 120 + 4
    boxes = torch.cat([c, c + wh], dim=1)
    scores = torch.rand(N, generator=gen)
    for thr in (0.3, 0.5):
        got = ops.nms(boxes.to(DEV), scores.to(DEV), thr).cpu()
        assert torch.equal(got, SD.nms(boxes, scores, thr)), (N, thr)
    assert ops.nms(boxes[:0].to(DEV), scores[:0].to(DEV), 0.5).numel() == 0


def test_maskgeneration_box_logic():
    """tools/tool.py:166-269 after the detector: target filtering, NMS, SAM box prompts, mask modes and early-outs."""
    from PIL import Image
    from anyedit_amd.segment_anything import SamPredictor
    from anyedit_amd.tools.tool import maskgeneration, select_target_boxes, boxes_to_pixels_xyxy
    torch.manual_seed(31)
    sam = tiny_sam()
    gen = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for p in sam.parameters():
            if p.abs().sum() == 0:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
        sam.mask_decoder.iou_token.weight.mul_(0.2)
    pred = SamPredictor(sam.to(DEV))
    img = Image.fromarray(torch.randint(0, 256, (300, 400, 3), generator=gen, dtype=torch.uint8).numpy())
    dets = (torch.tensor([[0.30, 0.40, 0.30, 0.40], [0.31, 0.41, 0.30, 0.40], [0.75, 0.70, 0.20, 0.30], [0.5, 0.5, 0.9, 0.9]]),
            ["cat(0.81)", "cat(0.62)", "cat(0.55)", "red sofa(0.90)"])
    det = lambda image_pil, prompt, bt, tt: (dets[0].clone(), list(dets[1]))

    px = boxes_to_pixels_xyxy(dets[0], 512, 512)
    assert torch.allclose(px[0], torch.tensor([76.8, 102.4, 230.4, 307.2]))
    b, s = select_target_boxes(px, dets[1], "cat")
    assert b.shape == (3, 4) and s.tolist() == pytest.approx([0.81, 0.62, 0.55])
    assert select_target_boxes(px, dets[1], "the sofa")[0].shape == (1, 4)                # word-overlap fallback
    assert select_target_boxes(px, dets[1], ["dog", "sofa"])[0].shape == (1, 4) and select_target_boxes(px, dets[1], "dog") is None

    masks, image_pil, none, union = maskgeneration(det, pred, img, "cat", mask_mode="count", target_object="cat", device=DEV)
    assert masks.shape == (2, 1, 512, 512) and masks.dtype == torch.bool and none is None        # NMS removed the duplicate cat
    assert union == pytest.approx((153.6 / 512) * (204.8 / 512), rel=1e-5) and image_pil.size == (512, 512)
    ref_masks, _, _ = pred.predict_torch(None, None, boxes=pred.transform.apply_boxes_torch(px[[0, 2]], (512, 512)).to(DEV),
                                         multimask_output=False)
    assert torch.equal(masks, ref_masks)

    mask_pil, _, bbox_pil, _ = maskgeneration(det, pred, img, "cat", mask_mode="merge", target_object="cat", device=DEV)
    merged = torch.from_numpy(np.array(mask_pil))
    assert torch.equal(merged, (ref_masks.sum(dim=0) > 0)[0].cpu())
    bb = np.array(bbox_pil)
    assert bb[150, 100] == 255 and bb[400, 400] == 255 and bb[10, 10] == 0
    mask_pil, _, bbox_pil, _ = maskgeneration(det, pred, img, "cat", mask_mode="max", target_object="cat", device=DEV)
    assert torch.equal(torch.from_numpy(np.array(mask_pil)), ref_masks[0, 0].cpu()) and np.array(bbox_pil)[400, 400] == 0
    assert maskgeneration(det, pred, img, "dog", target_object="dog", device=DEV)[0] is None
    none_det = lambda image_pil, prompt, bt, tt: (torch.zeros(0, 4), [])
    assert maskgeneration(none_det, pred, img, "cat", device=DEV)[0] is None
    out = maskgeneration(det, pred, img, "cat", mask_mode="count", target_object=None, device=DEV)
    assert out[0].shape[0] == 4                                                                   # no target -> no filtering, no NMS
