"""State-dict skeleton for the oracle (TEST INFRASTRUCTURE ONLY, see papr_oracle.py).

Builds zero tensors with the reference's state-dict names and shapes from a config dict
(pinned by tests/golden/g8_manifest.json), so the oracle can be driven
without the reference and without the product package.
"""
import torch

from .papr_oracle import embed_dims


def _mlp_shapes(prefix, d_in, ecfg):
    shapes = {}
    n, width, d_out = ecfg["n_ff_layer"], ecfg["d_ff"], ecfg["d_ff_out"]
    skips = ecfg.get("skip_layers", [])
    halves = ecfg.get("half_layers", []) or []
    for i in range(n):
        fan_in = d_in if i == 0 else width
        fan_out = d_out if i == n - 1 else width
        if i + 1 in halves:                     # models/mlp.py:27-30
            fan_out //= 2
        if i in halves:
            fan_in //= 2
        if i in skips:
            fan_in += d_in
        shapes["%s.mlp.model.%d.bias" % (prefix, 2 * i + 1)] = (fan_out,)
        if ecfg.get("use_wn", False):           # weight_norm(nn.Linear): weight_g (out, 1), weight_v (out, in) in place of weight (models/mlp.py:21,35-36)
            shapes["%s.mlp.model.%d.weight_g" % (prefix, 2 * i + 1)] = (fan_out, 1)
            shapes["%s.mlp.model.%d.weight_v" % (prefix, 2 * i + 1)] = (fan_out, fan_in)
        else:
            shapes["%s.mlp.model.%d.weight" % (prefix, 2 * i + 1)] = (fan_out, fan_in)
    if ecfg["norm"] == "layernorm":
        shapes[prefix + ".innorm.a_2"] = (d_in,)
        shapes[prefix + ".innorm.b_2"] = (d_in,)
        shapes[prefix + ".outnorm.a_2"] = (d_out,)
        shapes[prefix + ".outnorm.b_2"] = (d_out,)
    return shapes


def state_shapes(cfg, num_points):
    e = cfg["models"]["attn"]["embed"]
    dk, dq, dv = embed_dims(cfg)
    d_model = cfg["models"]["attn"]["d_model"]
    shapes = {"points": (num_points, 3), "points_influ_scores": (num_points, 1), "bkg_feats": (1, 3),
              "pc_feats": (num_points, cfg["geoms"]["point_feats"]["dim"])}
    pre = "proximity_attn.embed."
    shapes.update(_mlp_shapes(pre + "embed_k", dk, e["key"]))
    shapes.update(_mlp_shapes(pre + "embed_q", dq, e["query"]))
    shapes.update(_mlp_shapes(pre + "embed_v", dv, e["value"]))
    al = "proximity_attn.attention_layer."
    shapes[al + "w_k.weight"] = (d_model, e["key"]["d_ff_out"])
    shapes[al + "w_k.bias"] = (d_model,)
    shapes[al + "w_q.weight"] = (d_model, e["query"]["d_ff_out"])
    shapes[al + "w_q.bias"] = (d_model,)
    if cfg["models"]["use_renderer"]:
        c_in = e["value"]["d_ff_out"]
        r = "renderer."
        for name, (co, ci, ks) in {
            "inc.double_conv.0": (128, c_in, 3),
            "down1.maxpool_conv.1.double_conv.0": (256, 128, 3),
            "down2.maxpool_conv.1.double_conv.0": (512, 256, 3),
            "up1.conv.double_conv.0": (256, 512, 3),
            "up2.conv.double_conv.0": (128, 256, 3),
            "outc.conv": (3, 128, 1),
        }.items():
            shapes[r + name + ".weight"] = (co, ci, ks, ks)
            shapes[r + name + ".bias"] = (co,)
        shapes[r + "up1.up.weight"] = (512, 256, 2, 2)
        shapes[r + "up1.up.bias"] = (256,)
        shapes[r + "up2.up.weight"] = (256, 128, 2, 2)
        shapes[r + "up2.up.bias"] = (128,)
    return shapes


def empty_state(cfg, num_points, dtype=torch.float32):
    st = {k: torch.zeros(s, dtype=dtype) for k, s in state_shapes(cfg, num_points).items()}
    st["bkg_feats"] = torch.tensor([cfg["geoms"]["background"]["init_color"]], dtype=dtype)
    st["select_k"] = torch.tensor(cfg["geoms"]["points"]["select_k"], dtype=torch.int32)
    return st
