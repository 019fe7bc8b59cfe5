"""Pinning aids (TEST INFRASTRUCTURE): state-dict converters from this repo's restatements
of the un-vendored mmdet / mmcv layers to HuggingFace `transformers`' independent
implementations of the same published models, so that the [3P] arithmetic the oracle
restates from memory can be checked against code written by somebody else.

  pixel_decoder_to_hf   oracle/layers.py MSDeformAttnPixelDecoder (mmdet 2.25.1 names:
                        input_convs / encoder.layers.N.{attentions.0,norms,ffns} /
                        level_encoding / lateral_convs / output_convs / mask_feature)
                        -> transformers Mask2FormerPixelDecoder
  deformable_detr_to_hf oracle/deformable_detr.py ChannelMapper + DeformableDetrTransformer and
                        the class / box branches of oracle/bbox_head.py
                        -> transformers DeformableDetrForObjectDetection
  (oracle/swin.py::to_hf_state does the same for the Swin backbone.)

Only tests/ imports this.
"""


def pixel_decoder_to_hf(sd):
    out = {}
    for k, v in sd.items():
        leaf = k.split(".")[-1]
        if k.startswith("input_convs."):
            i, part = k.split(".")[1:3]
            out["input_projections.%s.%d.%s" % (i, 0 if part == "conv" else 1, leaf)] = v
        elif k == "level_encoding.weight":
            out["level_embed"] = v
        elif k.startswith("lateral_convs.0.") or k.startswith("output_convs.0."):
            part = k.split(".")[2]
            name = "adapter_1" if k.startswith("lateral") else "layer_1"
            out["%s.%d.%s" % (name, 0 if part == "conv" else 1, leaf)] = v
        elif k.startswith("mask_feature."):
            out["mask_projection." + leaf] = v
        elif k.startswith("encoder.layers."):
            n = k.split(".")[2]
            rest = ".".join(k.split(".")[3:])
            p = "encoder.layers.%s." % n
            if rest.startswith("attentions.0."):
                out[p + "self_attn." + rest[len("attentions.0."):]] = v
            elif rest.startswith("norms.0."):
                out[p + "self_attn_layer_norm." + leaf] = v
            elif rest.startswith("norms.1."):
                out[p + "final_layer_norm." + leaf] = v
            elif rest.startswith("ffns.0.layers.0.0."):
                out[p + "fc1." + leaf] = v
            elif rest.startswith("ffns.0.layers.1."):
                out[p + "fc2." + leaf] = v
            else:
                raise KeyError(k)
        else:
            raise KeyError(k)
    return out


def decoder_layer_to_hf(sd):
    """One masked-attention decoder layer (mmcv BaseTransformerLayer with operation order
    cross_attn, norm, self_attn, norm, ffn, norm; oracle/layers.py) ->
    transformers Mask2FormerMaskedAttentionDecoderLayer (whose self-attention is its own
    q/k/v-projection implementation, not nn.MultiheadAttention)."""
    out = {}
    for k, v in sd.items():
        leaf = k.split(".")[-1]
        if k.startswith("attentions.0.attn."):
            out["cross_attn." + k[len("attentions.0.attn."):]] = v
        elif k.startswith("attentions.1.attn.in_proj_"):
            for name, part in zip(("q_proj", "k_proj", "v_proj"), v.chunk(3, 0)):
                out["self_attn.%s.%s" % (name, "weight" if leaf.endswith("weight") else "bias")] = part
        elif k.startswith("attentions.1.attn.out_proj."):
            out["self_attn.out_proj." + leaf] = v
        elif k.startswith("norms."):
            name = ("cross_attn_layer_norm", "self_attn_layer_norm", "final_layer_norm")[int(k.split(".")[1])]
            out["%s.%s" % (name, leaf)] = v
        elif k.startswith("ffns.0.layers.0.0."):
            out["fc1." + leaf] = v
        elif k.startswith("ffns.0.layers.1."):
            out["fc2." + leaf] = v
        else:
            raise KeyError(k)
    return out


def resnet50_to_hf(sd):
    """oracle/backbone.py (mmdet / torchvision names) -> transformers ResNetBackbone."""
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if parts[0] in ("conv1", "bn1"):
            out["embedder.embedder.%s.%s" % ("convolution" if parts[0] == "conv1" else "normalization",
                                             parts[-1])] = v
            continue
        stage, blk = int(parts[0][5:]) - 1, parts[1]
        base = "encoder.stages.%d.layers.%s." % (stage, blk)
        if parts[2] == "downsample":
            out[base + "shortcut.%s.%s" % ("convolution" if parts[3] == "0" else "normalization",
                                           parts[-1])] = v
        else:
            i = int(parts[2][-1]) - 1                       # conv1/bn1 -> layer.0, ...
            kind = "convolution" if parts[2].startswith("conv") else "normalization"
            out[base + "layer.%d.%s.%s" % (i, kind, parts[-1])] = v
    return out


def self_first_layer_to_hf(sd):
    """A decoder layer with operation order (self_attn, norm, cross_attn, norm, ffn, norm) --
    CrossHeadBaseline's relation decoder -- -> transformers DetrDecoderLayer, whose self- and
    cross-attention are both its own q/k/v-projection code (no nn.MultiheadAttention)."""
    out = {}
    for k, v in sd.items():
        leaf = k.split(".")[-1]
        for j, name in ((0, "self_attn"), (1, "encoder_attn")):
            pre = "attentions.%d.attn." % j
            if k.startswith(pre + "in_proj_"):
                for proj, part in zip(("q_proj", "k_proj", "v_proj"), v.chunk(3, 0)):
                    out["%s.%s.%s" % (name, proj, "weight" if leaf.endswith("weight") else "bias")] = part
            elif k.startswith(pre + "out_proj."):
                out["%s.o_proj.%s" % (name, leaf)] = v
        if k.startswith("norms."):
            name = ("self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm")[int(k.split(".")[1])]
            out["%s.%s" % (name, leaf)] = v
        elif k.startswith("ffns.0.layers.0.0."):
            out["mlp.fc1." + leaf] = v
        elif k.startswith("ffns.0.layers.1."):
            out["mlp.fc2." + leaf] = v
    return out


def deformable_detr_to_hf(neck_sd, head_sd):
    """oracle ChannelMapper (mmdet names `convs.i.{conv,gn}` / `extra_convs.j.{conv,gn}`) +
    the trunk part of OracleCrossHeadBBox (`transformer.*`, `cls_branches.*`,
    `reg_branches.*`; mmdet 2.25.1 DeformableDETRHead / DeformableDetrTransformer names)
    -> transformers DeformableDetrForObjectDetection (two_stage, with_box_refine), everything
    except its backbone.  HF's input projections carry a conv bias that mmcv's ConvModule
    (conv followed by a norm) does not have: it is set to zero."""
    import torch
    out = {}
    n_convs = len({k.split(".")[1] for k in neck_sd if k.startswith("convs.")})
    for k, v in neck_sd.items():
        kind, i, part, leaf = k.split(".")
        lvl = int(i) + (n_convs if kind == "extra_convs" else 0)
        out["model.input_proj.%d.%d.%s" % (lvl, 0 if part == "conv" else 1, leaf)] = v
        if part == "conv":
            out["model.input_proj.%d.0.bias" % lvl] = torch.zeros(v.shape[0])
    for k, v in head_sd.items():
        leaf = k.split(".")[-1]
        if k.startswith("cls_branches."):
            i = k.split(".")[1]
            out["class_embed.%s.%s" % (i, leaf)] = v
            out["model.decoder.class_embed.%s.%s" % (i, leaf)] = v
        elif k.startswith("reg_branches."):
            _, i, j, _ = k.split(".")
            name = "bbox_embed.%s.layers.%d.%s" % (i, int(j) // 2, leaf)
            out[name] = v
            out["model.decoder." + name] = v
        elif k == "transformer.level_embeds":
            out["model.level_embed"] = v
        elif k.split(".")[1] in ("enc_output", "enc_output_norm", "pos_trans", "pos_trans_norm"):
            out["model." + k[len("transformer."):]] = v
        elif k.startswith("transformer.encoder.layers."):
            n = k.split(".")[3]
            rest = ".".join(k.split(".")[4:])
            p = "model.encoder.layers.%s." % n
            if rest.startswith("attentions.0."):
                out[p + "self_attn." + rest[len("attentions.0."):]] = v
            elif rest.startswith("norms."):
                out[p + ("self_attn_layer_norm.", "final_layer_norm.")[int(rest.split(".")[1])] + leaf] = v
            elif rest.startswith("ffns.0.layers.0.0."):
                out[p + "mlp.fc1." + leaf] = v
            elif rest.startswith("ffns.0.layers.1."):
                out[p + "mlp.fc2." + leaf] = v
            else:
                raise KeyError(k)
        elif k.startswith("transformer.decoder.layers."):
            n = k.split(".")[3]
            rest = ".".join(k.split(".")[4:])
            p = "model.decoder.layers.%s." % n
            if rest.startswith("attentions.0.attn.in_proj_"):
                for proj, part in zip(("q_proj", "k_proj", "v_proj"), v.chunk(3, 0)):
                    out[p + "self_attn.%s.%s" % (proj, "weight" if leaf.endswith("weight") else "bias")] = part
            elif rest.startswith("attentions.0.attn.out_proj."):
                out[p + "self_attn.o_proj." + leaf] = v
            elif rest.startswith("attentions.1."):
                out[p + "encoder_attn." + rest[len("attentions.1."):]] = v
            elif rest.startswith("norms."):
                out[p + ("self_attn_layer_norm.", "encoder_attn_layer_norm.",
                         "final_layer_norm.")[int(rest.split(".")[1])] + leaf] = v
            elif rest.startswith("ffns.0.layers.0.0."):
                out[p + "mlp.fc1." + leaf] = v
            elif rest.startswith("ffns.0.layers.1."):
                out[p + "mlp.fc2." + leaf] = v
            else:
                raise KeyError(k)
    return out
