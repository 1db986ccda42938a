"""CPU oracle for the PAPR per-ray render path.  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU restatement of the reference algorithm
(zvict/papr).  It is never imported by the product package ``papr_amd``;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may use it, and there only as the checker / reported baseline.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
itself (pure Python) in the build container and stores its outputs for fixed
seeded inputs under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors.

Everything is written functionally over a ``state`` dict that uses the
reference's state-dict key names, so the same tensors
drive the reference, this oracle and the HIP path.

Reference lines followed (relative to /root/reference):
  knn_select          models/model.py:258-283   (_calculate_global_distances)
  ray_geometry        models/model.py:285-310   (_calculate_distances), models/utils.py:255-257
  posenc              models/utils.py:232-242
  custom_layernorm    models/attn.py:30-42
  mlp_apply           models/mlp.py:12-59, models/utils.py:183-190
  embed_kqv           models/attn.py:165-197, models/model.py:396-437
  attention_scores    models/attn.py:212-226, :45-54
  attention_tail      models/model.py:519-545 / :473-492
  small_unet          models/unet.py:182-258
  render              models/model.py:494-560
  amp (render(..., amp=True))   models/attn.py:248, models/unet.py:212: the attention block and the U-Net under fp16 autocast
                      (`use_amp: true`, configs/default.yml:6-7) -- torch's CPU autocast, which casts the same ops of these regions as the
                      CUDA list does (linear / matmul / conv2d / conv_transpose2d; tests/golden/make_golden.py, shim 4); pinned by G17
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# configuration helpers (a cfg is the deep-merged YAML as a plain dict)
# --------------------------------------------------------------------------


def embed_dims(cfg):
    """Input widths of the key / query / value embedding MLPs (models/attn.py:138-150)."""
    e = cfg["models"]["attn"]["embed"]
    self_w = 1 if e["embed_type"] == 1 else 0
    pf = cfg["geoms"]["point_feats"]
    fdim = pf["dim"]
    dk = sum(3 * (self_w + 2 * L) for L in e["k_L"]) + (fdim if pf["use_ink"] else 0)
    dq = sum(3 * (self_w + 2 * L) for L in e["q_L"]) + (fdim if pf["use_inq"] else 0)
    dv = sum(3 * (self_w + 2 * L) for L in e["v_L"]) + (fdim if pf["use_inv"] else 0)
    return dk, dq, dv


# --------------------------------------------------------------------------
# stage 1: k nearest points to each ray
# --------------------------------------------------------------------------


def ray_point_distance(points, rays_o, rays_d, eps):
    """Perpendicular distance of every point to every ray, reference op order.

    points (P,3); rays_o (N,3); rays_d (N,H,W,3)  ->  (N,H,W,P)
    The direction is used as given (not re-normalised), models/model.py:276-279.
    """
    N = rays_d.shape[0]
    d = rays_d.unsqueeze(-2)                      # (N,H,W,1,3)
    o = rays_o.reshape(N, 1, 1, 1, 3)
    v = points.reshape(1, 1, 1, -1, 3) - o        # (N,1,1,P,3)
    t = torch.sum(v * d, dim=-1) / (torch.sum(d * d, dim=-1) + eps)
    perp = v - d * t.unsqueeze(-1)
    return torch.norm(perp, dim=-1)


def knn_select(points, rays_o, rays_d, k, eps, rows_per_chunk=8):
    """Indices (sorted by distance) and distances of the k nearest points per ray.

    The reference asks for an unsorted top-k (models/model.py:281); the set is what
    matters, so the oracle returns it sorted to make comparisons canonical.
    Returns idx (N,H,W,k) int64, dist (N,H,W,k).
    """
    N, H, W, _ = rays_d.shape
    idx_rows, dist_rows = [], []
    for h0 in range(0, H, rows_per_chunk):
        feat = ray_point_distance(points, rays_o, rays_d[:, h0:h0 + rows_per_chunk], eps)
        dist, idx = feat.topk(k, dim=-1, largest=False, sorted=True)
        idx_rows.append(idx)
        dist_rows.append(dist)
    return torch.cat(idx_rows, dim=1), torch.cat(dist_rows, dim=1)


# --------------------------------------------------------------------------
# stage 2: per-(ray, point) geometry and encodings
# --------------------------------------------------------------------------


def ray_geometry(sel_points, rays_o, rays_d, eps):
    """sel_points (N,H,W,k,3) -> (s, u): along-ray and perpendicular components of p - o.

    Uses the re-normalised direction r = d / (|d| + eps), models/model.py:302-305.
    """
    N = rays_d.shape[0]
    r = (rays_d / (torch.norm(rays_d, dim=-1, keepdim=True) + eps)).unsqueeze(-2)
    v = sel_points - rays_o.reshape(N, 1, 1, 1, 3)
    t = torch.sum(v * r, dim=-1) / (torch.sum(r * r, dim=-1) + eps)
    s = r * t.unsqueeze(-1)
    u = v - s
    return s, u


def posenc(x, L, factor=2.0, mult=1.0, with_self=True):
    """[x, sin(f^0 x m), cos(f^0 x m), sin(f^1 x m), ...] interleaved per component.

    (...,C) -> (...,C*(with_self + 2L)); models/utils.py:232-242.
    """
    parts = [x] if with_self else []
    for i in range(L):
        arg = factor ** i * x * mult
        parts.append(torch.sin(arg))
        parts.append(torch.cos(arg))
    return torch.stack(parts, dim=-1).flatten(-2, -1)


def custom_layernorm(x, gain, bias, eps):
    """a * (x - mean) / (std_unbiased + eps) + b ; models/attn.py:39-42."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return gain * (x - mean) / (std + eps) + bias


def _act(x, kind):
    kind = kind.lower()
    if kind == "none":
        return x
    if kind == "relu":
        return torch.relu(x)
    if kind == "leakyrelu":
        return F.leaky_relu(x, 0.2)
    raise NotImplementedError("oracle activation '%s'" % kind)


def mlp_apply(x, state, prefix, n_layer, act, last_act, skip_layers=()):
    """Linear/activation stack with optional re-concatenation of the input before layer i.

    Weights live at ``{prefix}.model.{2i+1}.{weight,bias}``; models/mlp.py:47-59.
    """
    x0 = x
    for i in range(n_layer):
        if i in skip_layers:
            x = torch.cat([x, x0], dim=-1)
        gkey = "%s.model.%d.weight_g" % (prefix, 2 * i + 1)
        if gkey in state:                       # weight_norm(nn.Linear) (models/mlp.py:21,35-36): W = g v / |v|, the norm taken row by row
            v = state["%s.model.%d.weight_v" % (prefix, 2 * i + 1)]
            w = state[gkey] * v / v.norm(dim=1, keepdim=True)
        else:
            w = state["%s.model.%d.weight" % (prefix, 2 * i + 1)]
        b = state["%s.model.%d.bias" % (prefix, 2 * i + 1)]
        x = F.linear(x, w, b)
        x = _act(x, last_act if i == n_layer - 1 else act)
    return x


def feed_forward(x, state, prefix, ecfg, eps):
    """outnorm(mlp(innorm(x))); dropout p=0 and residual off in every shipped config."""
    if ecfg.get("residual_ff", False) and x.shape[-1] == ecfg["d_ff_out"]:
        raise NotImplementedError("residual_ff")
    normed = ecfg["norm"] == "layernorm"
    if normed:
        x = custom_layernorm(x, state[prefix + ".innorm.a_2"], state[prefix + ".innorm.b_2"], eps)
    x = mlp_apply(x, state, prefix + ".mlp", ecfg["n_ff_layer"], ecfg["ff_act"],
                  ecfg["ff_last_act"], tuple(ecfg.get("skip_layers", [])))
    if normed:
        x = custom_layernorm(x, state[prefix + ".outnorm.a_2"], state[prefix + ".outnorm.b_2"], eps)
    return x


def build_inputs(state, cfg, rays_o, rays_d, idx):
    """Raw (pre-LayerNorm) key / query / value input rows for given neighbour indices."""
    eps = cfg["eps"]
    e = cfg["models"]["attn"]["embed"]
    pf = cfg["geoms"]["point_feats"]
    with_self = e["embed_type"] == 1
    pe = lambda x, L: posenc(x, L, e["pe_factor"], e["pe_mult_factor"], with_self)
    sel = state["points"][idx]                                  # (N,H,W,k,3)
    s, u = ray_geometry(sel, rays_o, rays_d, eps)
    k_parts = [pe(sel.detach(), e["k_L"][0]), pe(s, e["k_L"][1]), pe(u, e["k_L"][2])]
    q_parts = [pe(rays_d.unsqueeze(-2), e["q_L"][0])]
    v_parts = [pe(s, e["v_L"][0]), pe(u, e["v_L"][1])]
    feats = state["pc_feats"][idx] if (pf["use_ink"] or pf["use_inq"] or pf["use_inv"]) else None
    if pf["use_ink"]:
        k_parts.append(feats)
    if pf["use_inq"]:
        q_parts.append(feats)
    if pf["use_inv"]:
        v_parts.append(feats)
    key_in = torch.cat(k_parts, dim=-1).flatten(0, 2)           # (R,k,dk)
    qry_in = torch.cat(q_parts, dim=-1).flatten(0, 2)           # (R,1,dq)
    val_in = torch.cat(v_parts, dim=-1).flatten(0, 2)           # (R,k,dv)
    return key_in, qry_in, val_in, sel, s, u


def embed_kqv(state, cfg, key_in, qry_in, val_in):
    e = cfg["models"]["attn"]["embed"]
    eps = cfg["eps"]
    pre = "proximity_attn.embed."
    K = feed_forward(key_in, state, pre + "embed_k", e["key"], eps)
    Q = feed_forward(qry_in, state, pre + "embed_q", e["query"], eps)
    V = feed_forward(val_in, state, pre + "embed_v", e["value"], eps)
    return K, Q, V


def attention_scores(state, cfg, K, Q):
    """relu( (W_q Q + b_q) . (W_k K_j + b_k) / sqrt(d_model) ) -> (R,k)."""
    pre = "proximity_attn.attention_layer."
    kp = F.linear(K, state[pre + "w_k.weight"], state[pre + "w_k.bias"])    # (R,k,d)
    qp = F.linear(Q, state[pre + "w_q.weight"], state[pre + "w_q.bias"])    # (R,1,d)
    d_model = cfg["models"]["attn"]["d_model"]
    sc = torch.matmul(qp, kp.transpose(-2, -1)) / math.sqrt(d_model)        # (R,1,k)
    return _act(sc, cfg["models"]["attn"]["score_act"]).squeeze(1)


def attention_tail(scores, influ, V, bkg_score, normalize):
    """scores (R,k), influ (R,k), V (R,k,C) -> fused (R,C), attn (R,k+1).

    z = [e_j * w_j ..., B]; a = softmax(z); fused = sum_j a_j/sum(a_1..k) * V_j.
    """
    z = torch.cat([scores * influ, torch.full_like(scores[:, :1], float(bkg_score))], dim=-1)
    a = torch.softmax(z, dim=-1)
    top = a[:, :-1]
    if normalize:
        top = top / torch.sum(top, dim=-1, keepdim=True)
    fused = torch.sum(V * top.unsqueeze(-1), dim=1)
    return fused, a


def small_unet(state, x, prefix="renderer."):
    """32->128 ; pool,128->256 ; pool,256->512 ; up+cat,512->256 ; up+cat,256->128 ; 1x1 -> 3."""
    g = lambda n: state[prefix + n]
    c3 = lambda t, n: torch.relu(F.conv2d(t, g(n + ".weight"), g(n + ".bias"), padding=1))
    x1 = c3(x, "inc.double_conv.0")
    x2 = c3(F.max_pool2d(x1, 2), "down1.maxpool_conv.1.double_conv.0")
    x3 = c3(F.max_pool2d(x2, 2), "down2.maxpool_conv.1.double_conv.0")

    def up(lo, skip, name):
        lo = F.conv_transpose2d(lo, g(name + ".up.weight"), g(name + ".up.bias"), stride=2)
        dy, dx = skip.shape[2] - lo.shape[2], skip.shape[3] - lo.shape[3]
        lo = F.pad(lo, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return c3(torch.cat([skip, lo], dim=1), name + ".conv.double_conv.0")

    y = up(x3, x2, "up1")
    y = up(y, x1, "up2")
    return F.conv2d(y, g("outc.conv.weight"), g("outc.conv.bias"))


def _autocast(amp, dtype=torch.float16):
    return torch.autocast("cpu", dtype=dtype, enabled=bool(amp))


def render(state, cfg, rays_o, rays_d, idx=None, want_rgb=True, amp=False):
    """Full per-ray path.  Returns a dict of every intermediate the parity tests look at.
    amp: the reference's `use_amp: true` arithmetic (fp16 autocast around the attention block and the U-Net; everything else fp32)."""
    N, H, W, _ = rays_d.shape
    pts = state["points"]
    k = int(cfg["geoms"]["points"]["select_k"])
    out = {}
    if idx is None:
        if k >= pts.shape[0] or k < 0:
            idx = torch.arange(pts.shape[0]).expand(N, H, W, -1)
        else:
            with torch.no_grad():
                idx, dist = knn_select(pts.detach(), rays_o, rays_d, k, cfg["eps"])
            out["knn_dist"] = dist
    out["idx"] = idx
    key_in, qry_in, val_in, sel, s, u = build_inputs(state, cfg, rays_o, rays_d, idx)
    out.update(key_in=key_in, qry_in=qry_in, val_in=val_in, sel_points=sel, s=s, u=u)
    with _autocast(amp):                            # models/attn.py:248
        K, Q, V = embed_kqv(state, cfg, key_in, qry_in, val_in)
        scores = attention_scores(state, cfg, K, Q)
    out.update(K=K, Q=Q, V=V, scores=scores)
    influ = state["points_influ_scores"][idx].reshape(-1, idx.shape[-1])
    fused, attn = attention_tail(scores, influ, V, cfg["geoms"]["background"]["constant"],
                                 cfg["models"]["normalize_topk_attn"])
    C = fused.shape[-1]
    out["fused"] = fused.reshape(N, H, W, C)
    out["attn"] = attn.reshape(N, H, W, -1)
    if not want_rgb:
        return out
    fmap = out["fused"]
    if cfg["models"]["use_renderer"]:
        with _autocast(amp):                        # models/unet.py:212
            fg = small_unet(state, fmap.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    else:
        fg = fmap
    a_bkg = out["attn"][..., -1:]
    bkg = state["bkg_feats"].reshape(1, 1, 1, 3)
    if cfg["models"]["normalize_topk_attn"]:
        rgb = fg * (1 - a_bkg) + bkg * a_bkg
    else:
        rgb = fg + bkg * a_bkg
    out["rgb"] = rgb
    return out


# --------------------------------------------------------------------------
# training step of the oracle (cpu_baseline and gradient parity)
# --------------------------------------------------------------------------

PER_POINT = ("points", "points_influ_scores", "pc_feats")


def trainable_state(state, cfg):
    """Clone a state dict into leaf tensors; everything but bkg_feats/select_k gets a gradient."""
    out = {}
    for name, t in state.items():
        t = t.detach().clone()
        if t.is_floating_point() and not (name == "bkg_feats" and not cfg["geoms"]["background"]["learnable"]):
            t.requires_grad_(True)
        out[name] = t
    return out


def param_groups(state):
    """The reference's optimizer grouping (models/model.py:117-167)."""
    groups = {"points": [], "attn": [], "points_influ_scores": [], "pc_feats": [], "renderer": []}
    for name, t in state.items():
        if not t.requires_grad:
            continue
        if name in PER_POINT:
            groups[name].append(t)
        elif name.startswith("proximity_attn."):
            groups["attn"].append(t)
        elif name.startswith("renderer."):
            groups["renderer"].append(t)
    return {k: v for k, v in groups.items() if v}


def make_optimizers(state, cfg):
    lr = cfg["training"]["lr"]
    key = {"points": "points", "attn": "attn", "points_influ_scores": "points_influ_scores",
           "pc_feats": "feats", "renderer": "generator"}
    opts = {}
    for g, params in param_groups(state).items():
        o = lr[key[g]]
        opts[g] = torch.optim.Adam(params, lr=o["base_lr"] * lr["lr_factor"],
                                   weight_decay=o.get("weight_decay", 0))
    return opts


def train_step(state, opts, cfg, rays_o, rays_d, target):
    """clear grads, forward, MSE, backward, Adam step for every group; returns loss (float)."""
    for o in opts.values():
        o.zero_grad()
    rgb = render(state, cfg, rays_o, rays_d)["rgb"]
    loss = torch.mean((rgb - target) ** 2)
    loss.backward()
    for o in opts.values():
        o.step()
    return float(loss.detach())
