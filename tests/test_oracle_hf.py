"""Pin the oracle against two independent offline implementations (HF
`transformers`): the CLIP text tower and the ViT block / DPT neck.  The
reference itself ships no golden vectors, so these cross-checks
are what anchors oracle/lseg_oracle.py beyond line-by-line correspondence."""
import pytest
import torch

from lseg_hip.config import get_config
from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images
from oracle import lseg_oracle as O

transformers = pytest.importorskip("transformers")


def _copy(dst, src):
    assert dst.shape == src.shape, (dst.shape, src.shape)
    with torch.no_grad():
        dst.copy_(src.float())


def test_text_tower_matches_hf_clip():
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    cfg = get_config("tiny16")
    t = cfg.text
    sd = synthetic_state_dict(cfg, seed=3, clip_fp16=False)
    hf_cfg = CLIPTextConfig(vocab_size=t.vocab, hidden_size=t.width, intermediate_size=4 * t.width,
                            projection_dim=t.embed_dim, num_hidden_layers=t.layers,
                            num_attention_heads=t.heads, max_position_embeddings=t.ctx,
                            hidden_act="quick_gelu", layer_norm_eps=1e-5,
                            eos_token_id=t.vocab - 1, bos_token_id=t.vocab - 2, pad_token_id=0)
    m = CLIPTextModelWithProjection(hf_cfg).eval()
    cp = "clip_pretrained."
    tm = m.text_model
    _copy(tm.embeddings.token_embedding.weight, sd[cp + "token_embedding.weight"])
    _copy(tm.embeddings.position_embedding.weight, sd[cp + "positional_embedding"])
    W = t.width
    for i, lyr in enumerate(tm.encoder.layers):
        b = f"{cp}transformer.resblocks.{i}."
        ipw, ipb = sd[b + "attn.in_proj_weight"], sd[b + "attn.in_proj_bias"]
        for j, proj in enumerate((lyr.self_attn.q_proj, lyr.self_attn.k_proj, lyr.self_attn.v_proj)):
            _copy(proj.weight, ipw[j * W:(j + 1) * W]); _copy(proj.bias, ipb[j * W:(j + 1) * W])
        _copy(lyr.self_attn.out_proj.weight, sd[b + "attn.out_proj.weight"])
        _copy(lyr.self_attn.out_proj.bias, sd[b + "attn.out_proj.bias"])
        _copy(lyr.layer_norm1.weight, sd[b + "ln_1.weight"]); _copy(lyr.layer_norm1.bias, sd[b + "ln_1.bias"])
        _copy(lyr.layer_norm2.weight, sd[b + "ln_2.weight"]); _copy(lyr.layer_norm2.bias, sd[b + "ln_2.bias"])
        _copy(lyr.mlp.fc1.weight, sd[b + "mlp.c_fc.weight"]); _copy(lyr.mlp.fc1.bias, sd[b + "mlp.c_fc.bias"])
        _copy(lyr.mlp.fc2.weight, sd[b + "mlp.c_proj.weight"]); _copy(lyr.mlp.fc2.bias, sd[b + "mlp.c_proj.bias"])
    _copy(tm.final_layer_norm.weight, sd[cp + "ln_final.weight"])
    _copy(tm.final_layer_norm.bias, sd[cp + "ln_final.bias"])
    _copy(m.text_projection.weight, sd[cp + "text_projection"].t())
    tok = synthetic_tokens(["wall", "sky", "potted plant", "a very long label of many words"], t.vocab, t.ctx)
    with torch.no_grad():
        ref = m(input_ids=tok).text_embeds
        ours = O.encode_text(sd, tok, t, emulate_fp16=False)
    assert torch.allclose(ours, ref, atol=2e-5, rtol=1e-4), (ours - ref).abs().max()
    # the fp16-emulated tower stays close to the fp32 one (fp16 noise only)
    with torch.no_grad():
        sd16 = synthetic_state_dict(cfg, seed=3, clip_fp16=True)
        emu = O.encode_text(sd16, tok, t, emulate_fp16=True)
    assert (emu - ref).abs().max() < 3e-2 * ref.abs().max()


def test_vit_blocks_match_hf_vit():
    from transformers import ViTConfig, ViTModel
    cfg = get_config("tiny16")
    sd = synthetic_state_dict(cfg, seed=5)
    side = cfg.pos_grid * cfg.patch          # native grid -> pos-embed resize is the identity
    hf_cfg = ViTConfig(hidden_size=cfg.dim, num_hidden_layers=cfg.depth, num_attention_heads=cfg.heads,
                       intermediate_size=4 * cfg.dim, hidden_act="gelu", layer_norm_eps=1e-6,
                       image_size=side, patch_size=cfg.patch, qkv_bias=True,
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = ViTModel(hf_cfg, add_pooling_layer=False).eval()
    vm = "pretrained.model."
    _copy(m.embeddings.cls_token, sd[vm + "cls_token"])
    _copy(m.embeddings.position_embeddings, sd[vm + "pos_embed"])
    _copy(m.embeddings.patch_embeddings.projection.weight, sd[vm + "patch_embed.proj.weight"])
    _copy(m.embeddings.patch_embeddings.projection.bias, sd[vm + "patch_embed.proj.bias"])
    D = cfg.dim
    for i, lyr in enumerate(m.layers):
        b = f"{vm}blocks.{i}."
        att = lyr.attention
        for j, proj in enumerate((att.q_proj, att.k_proj, att.v_proj)):
            _copy(proj.weight, sd[b + "attn.qkv.weight"][j * D:(j + 1) * D])
            _copy(proj.bias, sd[b + "attn.qkv.bias"][j * D:(j + 1) * D])
        _copy(att.o_proj.weight, sd[b + "attn.proj.weight"])
        _copy(att.o_proj.bias, sd[b + "attn.proj.bias"])
        _copy(lyr.layernorm_before.weight, sd[b + "norm1.weight"]); _copy(lyr.layernorm_before.bias, sd[b + "norm1.bias"])
        _copy(lyr.layernorm_after.weight, sd[b + "norm2.weight"]); _copy(lyr.layernorm_after.bias, sd[b + "norm2.bias"])
        _copy(lyr.mlp.fc1.weight, sd[b + "mlp.fc1.weight"]); _copy(lyr.mlp.fc1.bias, sd[b + "mlp.fc1.bias"])
        _copy(lyr.mlp.fc2.weight, sd[b + "mlp.fc2.weight"]); _copy(lyr.mlp.fc2.bias, sd[b + "mlp.fc2.bias"])
    x = synthetic_images(2, side, side, seed=1)
    with torch.no_grad():
        hs = m(pixel_values=x, output_hidden_states=True).hidden_states   # [emb, blk0, blk1, ...]
        acts = O.forward_flex(sd, x, cfg)
    for a, hook in zip(acts, cfg.hooks):
        ref = hs[hook + 1]
        assert torch.allclose(a, ref, atol=5e-5, rtol=1e-4), (a - ref).abs().max()


def test_dpt_neck_matches_hf_dpt():
    from transformers import DPTConfig
    from transformers.models.dpt.modeling_dpt import DPTNeck
    cfg = get_config("tiny16")
    sd = synthetic_state_dict(cfg, seed=7)
    hf_cfg = DPTConfig(hidden_size=cfg.dim, neck_hidden_sizes=list(cfg.reassemble),
                       reassemble_factors=[4, 2, 1, 0.5], fusion_hidden_size=cfg.features,
                       readout_type="project", hidden_act="gelu", is_hybrid=False,
                       use_batch_norm_in_fusion_residual=True, use_bias_in_fusion_residual=False)
    neck = DPTNeck(hf_cfg).eval()
    for l in range(4):
        a = f"pretrained.act_postprocess{l + 1}."
        _copy(neck.reassemble_stage.readout_projects[l][0].weight, sd[a + "0.project.0.weight"])
        _copy(neck.reassemble_stage.readout_projects[l][0].bias, sd[a + "0.project.0.bias"])
        lay = neck.reassemble_stage.layers[l]
        _copy(lay.projection.weight, sd[a + "3.weight"]); _copy(lay.projection.bias, sd[a + "3.bias"])
        if cfg.resample[l][0] != "id":
            _copy(lay.resize.weight, sd[a + "4.weight"]); _copy(lay.resize.bias, sd[a + "4.bias"])
        _copy(neck.convs[l].weight, sd[f"scratch.layer{l + 1}_rn.weight"])
    for i, r in enumerate((4, 3, 2, 1)):          # HF fuses from the deepest level first
        p = f"scratch.refinenet{r}."
        fl = neck.fusion_stage.layers[i]
        _copy(fl.projection.weight, sd[p + "out_conv.weight"]); _copy(fl.projection.bias, sd[p + "out_conv.bias"])
        for u, rl in ((1, fl.residual_layer1), (2, fl.residual_layer2)):
            q = f"{p}resConfUnit{u}."
            _copy(rl.convolution1.weight, sd[q + "conv1.weight"]); _copy(rl.convolution2.weight, sd[q + "conv2.weight"])
            for c, bn in ((1, rl.batch_norm1), (2, rl.batch_norm2)):
                _copy(bn.weight, sd[q + f"bn{c}.weight"]); _copy(bn.bias, sd[q + f"bn{c}.bias"])
                _copy(bn.running_mean, sd[q + f"bn{c}.running_mean"]); _copy(bn.running_var, sd[q + f"bn{c}.running_var"])
    gh = gw = 6
    g = torch.Generator().manual_seed(11)
    acts = [torch.randn((2, 1 + gh * gw, cfg.dim), generator=g) for _ in range(4)]
    with torch.no_grad():
        ref = neck(acts, gh, gw)          # [path_4, path_3, path_2, path_1]
        layers = [O.act_postprocess(sd, l, acts[l], gh, gw, cfg) for l in range(4)]
        rn = [torch.nn.functional.conv2d(layers[l], sd[f"scratch.layer{l + 1}_rn.weight"], None, padding=1)
              for l in range(4)]
        p4 = O.fusion_block(sd, "scratch.refinenet4.", rn[3])
        p3 = O.fusion_block(sd, "scratch.refinenet3.", p4, rn[2])
        p2 = O.fusion_block(sd, "scratch.refinenet2.", p3, rn[1])
        p1 = O.fusion_block(sd, "scratch.refinenet1.", p2, rn[0])
    for ours, r in zip((p4, p3, p2, p1), ref):
        assert ours.shape == r.shape
        assert torch.allclose(ours, r, atol=1e-4, rtol=1e-4), (ours - r).abs().max()
