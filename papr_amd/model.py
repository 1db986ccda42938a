"""PAPR nn.Module on the HIP render path.

Same object protocol as the reference's `PAPR` (models/model.py:17-641; surface enumerated in
SURVEY.md section 8b): constructor arguments, parameter / buffer names, state-dict keys, optimizer
and scheduler dictionaries, `forward` / `evaluate` / `step` / `prune_points` / `add_points` /
`save` / `load` semantics.  What differs is where the arithmetic runs: selection, gather, geometry,
positional encoding, the three embedding MLPs, w_k / w_q and the attention tail execute in
libpapr_hip.so; only the U-Net render head, the compositing line and the optimizers stay in torch.
There is no CPU implementation of the render path in this package -- calling `forward` without a
ROCm device raises.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import adam as own_adam
from . import dist as pdist
from .activations import output_activation
from .config import as_node
from . import hip
from .ops import RenderPath, ln_fold_batch, ln_fold_job, prepare_mlp_weights, render_rays
from .pointcloud import grow_points, grow_points_device
from .schedule import create_learning_rate_fn, fast_forward
from .unet import get_generator

_OWN_ADAM = os.environ.get("PAPR_OWN_ADAM", "1") == "1"
_LN_FOLD_BATCH = os.environ.get("PAPR_LN_FOLD_BATCH", "1") == "1"      # (0: one papr_ln_fold launch per LayerNorm affine each way, A/B)
_OWN_ADAM_AMP = os.environ.get("PAPR_OWN_ADAM_AMP", "1") == "1"        # (0: torch's fused Adam through GradScaler.step under use_amp, A/B)
_OWN_COMPOSITE = os.environ.get("PAPR_OWN_COMPOSITE", "1") == "1"      # (0: the compositing line in torch ops, A/B)


def count_parameters(module):
    return sum(p.numel() for p in module.parameters() if p.requires_grad)


# ----------------------------------------------------------------------------------------------
# parameter containers with the reference's state-dict names (models/attn.py, models/mlp.py)
# ----------------------------------------------------------------------------------------------
class _NormParams(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))


class _MlpParams(nn.Module):
    """`model.{2i+1}` holds Linear i, matching the reference's [Identity, (Linear, act) x n] list."""

    def __init__(self, d_in, ecfg):
        super().__init__()
        n, width, d_out = ecfg["n_ff_layer"], ecfg["d_ff"], ecfg["d_ff_out"]
        skips = ecfg.get("skip_layers", []) or []
        halves = ecfg.get("half_layers", []) or []
        mods = [nn.Identity()]
        for i in range(n):
            fan_in, fan_out = (d_in if i == 0 else width), (d_out if i == n - 1 else width)
            if i + 1 in halves:                       # (models/mlp.py:27-30: the layer in front of a half layer produces half as much)
                fan_out //= 2
            if i in halves:
                fan_in //= 2
            if i in skips:
                fan_in += d_in
            lin = nn.Linear(fan_in, fan_out)
            if ecfg.get("use_wn", False):
                # `weight_norm(nn.Linear(..), name='weight')` (models/mlp.py:21,35-36): parameters weight_g (out, 1) and weight_v (out, in) in place
                # of weight, W = g v / |v| row by row.  torch's own (deprecated, still shipped) function: the same parameter names, order and
                # initial values; the kernels get W from `effective_weight` below
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    lin = torch.nn.utils.weight_norm(lin, name="weight")
            mods += [lin, nn.Identity()]
        self.model = nn.ModuleList(mods)
        for p in self.model.parameters():             # (weight_g is two-dimensional too: the reference re-draws it like a weight, mlp.py:43-45)
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def linears(self):
        return [m for m in self.model if isinstance(m, nn.Linear)]

    @staticmethod
    def effective_weight(lin):
        """The Linear's weight as the kernels take it: the parameter itself, or g v / |v| under weight norm (differentiable)."""
        if hasattr(lin, "weight_g"):
            return torch._weight_norm(lin.weight_v, lin.weight_g, 0)
        return lin.weight


class _FeedForwardParams(nn.Module):
    def __init__(self, d_in, ecfg):
        super().__init__()
        if ecfg["norm"] == "layernorm":
            self.innorm = _NormParams(d_in)
            self.outnorm = _NormParams(ecfg["d_ff_out"])
        else:
            self.innorm = nn.Identity()
            self.outnorm = nn.Identity()
        self.mlp = _MlpParams(d_in, ecfg)


class _EmbeddingParams(nn.Module):
    def __init__(self, plan, ecfg):
        super().__init__()
        self.embed_k = _FeedForwardParams(plan.key_w, ecfg["key"])
        self.embed_q = _FeedForwardParams(plan.qry_w, ecfg["query"])
        self.embed_v = _FeedForwardParams(plan.val_w, ecfg["value"])


class _AttentionLayerParams(nn.Module):
    def __init__(self, ecfg, d_model):
        super().__init__()
        self.w_k = nn.Linear(ecfg["key"]["d_ff_out"], d_model)
        self.w_q = nn.Linear(ecfg["query"]["d_ff_out"], d_model)
        nn.init.xavier_uniform_(self.w_k.weight)
        nn.init.xavier_uniform_(self.w_q.weight)


class ProximityAttentionParams(nn.Module):
    """Weights of the proximity-attention block (reference models/attn.py:229-252); no forward --
    the arithmetic is `papr_amd.ops.render_rays`."""

    def __init__(self, plan, acfg):
        super().__init__()
        self.embed = _EmbeddingParams(plan, acfg["embed"])
        self.attention_layer = _AttentionLayerParams(acfg["embed"], acfg["d_model"])

    def kernel_weights(self, plan):
        """Effective (LayerNorm-affine-folded, zero-padded) weights for the kernels, inside autograd.  Under `no_grad` (the chunk loops of
        test_step / eval_step: 64 calls per 800 x 800 image) they are prepared once and reused for as long as no parameter has been written
        (every in-place update -- an optimizer step, load_state_dict -- bumps the tensor's version counter; the writers that go around it --
        papr_adam_step, dist.broadcast_module_state -- bump dist.param_epoch)."""
        if not torch.is_grad_enabled():
            key = (pdist.param_epoch(),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
            cached = getattr(self, "_kw_cache", None)
            if cached is None or cached[0] != key:
                self._kw_cache = cached = (key, self._kernel_weights(plan))
            return cached[1]
        self._kw_cache = None
        return self._kernel_weights(plan)

    def _kernel_weights(self, plan):
        def ff(block, spec):
            lin = block.mlp.linears()
            ln = (block.innorm.a_2, block.innorm.b_2) if isinstance(block.innorm, _NormParams) else None
            return spec, [_MlpParams.effective_weight(l) for l in lin], [l.bias for l in lin], ln

        def proj(lin, block, spec):
            ln = (block.outnorm.a_2, block.outnorm.b_2) if isinstance(block.outnorm, _NormParams) else None
            return spec, [lin.weight], [lin.bias], ln

        e, a = self.embed, self.attention_layer
        todo = {"key": ff(e.embed_k, plan.key), "query": ff(e.embed_q, plan.qry), "value": ff(e.embed_v, plan.val),
                "wk": proj(a.w_k, e.embed_k, plan.wk), "wq": proj(a.w_q, e.embed_q, plan.wq)}
        # the LayerNorm affines folded into the Linear layers behind them (in front of key / query / value MLP, behind key / query into w_k / w_q): all
        # of them in ONE launch each way (four launches of 5 us each way before; PAPR_LN_FOLD_BATCH=0 for the A/B)
        folds = {n: ln_fold_job(*t) for n, t in todo.items()} if _LN_FOLD_BATCH else {}
        names = [n for n, f in folds.items() if f is not None]
        done = dict(zip(names, ln_fold_batch([folds[n] for n in names]))) if len(names) > 1 and len(names) <= hip.LN_FOLD_MAX_JOBS else {}
        out = {n: prepare_mlp_weights(*t, prefolded=done.get(n)) for n, t in todo.items()}
        if isinstance(e.embed_v.outnorm, _NormParams):           # (value.norm: layernorm -- applied by ops._RenderFn, nothing to fold it into)
            out["v_out"] = (e.embed_v.outnorm.a_2, e.embed_v.outnorm.b_2)
        return out


# ----------------------------------------------------------------------------------------------
class PAPR(nn.Module):
    def __init__(self, args, device="cuda"):
        super().__init__()
        args = as_node(args)
        self.args = args
        self.eps = args.eps
        self.device = device
        self.use_amp = args.use_amp
        self.amp_dtype = torch.float16 if args.amp_dtype == "float16" else torch.bfloat16
        self.scaler = torch.amp.GradScaler("cuda", enabled=bool(self.use_amp) and torch.cuda.is_available())

        popt, fopt, bopt = args.geoms.points, args.geoms.point_feats, args.geoms.background
        self.exposure_opt = args.exposure_control
        if self.exposure_opt.use:
            raise NotImplementedError("papr_amd: exposure-control fine-tuning (mapping MLP / cIMLE) is outside the render path built here")
        self.mapping_mlp = None
        self.plan = RenderPath(args)

        self.register_buffer("select_k", torch.tensor(popt.select_k, device=device, dtype=torch.int32))
        self.coord_scale = args.dataset.coord_scale

        # point positions: file, fibonacci sphere, or lattice cube (reference model.py:39-59, 194-256)
        if popt.load_path:
            pts = np.asarray(torch.load(popt.load_path, map_location="cpu")).astype(np.float32)
            np.random.shuffle(pts)
            points = torch.from_numpy(pts[:args.max_num_pts, :]).float()
        else:
            center = [c * self.coord_scale for c in popt.init_center]
            scale = [s * self.coord_scale for s in popt.init_scale]
            if popt.init_type == "sphere":
                points = self._sphere_pc(center, popt.init_num, scale)
            elif popt.init_type == "cube":
                points = self._cube_normal_pc(center, popt.init_num, scale)
            else:
                raise NotImplementedError("Point init type [{:s}] is not found".format(popt.init_type))
        self.points = nn.Parameter(points, requires_grad=True)
        self.points_influ_scores = nn.Parameter(torch.ones(points.shape[0], 1, device=device) * popt.influ_init_val)

        if args.models.use_renderer:
            self.renderer = get_generator(args.models.renderer.generator, in_c=args.models.attn.embed.value.d_ff_out,
                                          out_c=3, use_amp=self.use_amp, amp_dtype=self.amp_dtype)
        else:
            assert args.models.attn.embed.value.d_ff_out == 3, \
                "Value embedding MLP should have output dim 3 if not using renderer"

        self.bkg_feats = nn.Parameter(torch.FloatTensor(bopt.init_color)[None, :], requires_grad=bopt.learnable)
        self.bkg_score = torch.tensor(bopt.constant, device=device, dtype=torch.float32).reshape(1)

        self.use_pc_feats = fopt.use_ink or fopt.use_inq or fopt.use_inv
        if self.use_pc_feats:
            self.pc_feats = nn.Parameter(torch.randn(points.shape[0], fopt.dim), requires_grad=True)

        self.last_act = output_activation(args.models.last_act)
        self.proximity_attn = ProximityAttentionParams(self.plan, args.models.attn)
        self.added_points = False
        self.attn_lr = self.pts_lr = 0
        self.init_optimizers(total_steps=0)

    # ------------------------------------------------------------------------------- point clouds
    @staticmethod
    def _sphere_pc(center, num_pts, scale):
        i = np.arange(num_pts, dtype=np.float64)
        y = 1 - (i / float(num_pts - 1)) * 2
        rad = np.sqrt(1 - y * y)
        theta = math.pi * (3.0 - math.sqrt(5.0)) * i
        pts = np.stack([np.cos(theta) * rad * scale[0] + center[0], y * scale[1] + center[1],
                        np.sin(theta) * rad * scale[2] + center[2]], axis=-1)
        return torch.from_numpy(pts).float()

    @staticmethod
    def _cube_normal_pc(center, num_pts, scale):
        n_axis = int(num_pts ** (1.0 / 3.0))
        axes = [np.linspace(-scale[a], scale[a], n_axis) + center[a] for a in range(3)]
        gx, gy, gz = np.meshgrid(*axes, indexing="ij")
        pts = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=-1)
        rest = num_pts - pts.shape[0]
        if rest > 0:
            extra = np.stack([np.random.uniform(-scale[a], scale[a], rest) + center[a] for a in range(3)], axis=-1)
            pts = np.concatenate([pts, extra], axis=0)
        return torch.from_numpy(pts).float()

    # --------------------------------------------------------------------------------- optimizers
    def _groups(self):
        lr = self.args.training.lr
        groups = [("points", [self.points], lr.points), ("attn", list(self.proximity_attn.parameters()), lr.attn),
                  ("points_influ_scores", [self.points_influ_scores], lr.points_influ_scores)]
        if self.use_pc_feats:
            groups.append(("pc_feats", [self.pc_feats], lr.feats))
        if self.mapping_mlp is not None:
            groups.append(("mapping_mlp", list(self.mapping_mlp.parameters()), lr.mapping_mlp))
        if self.args.models.use_renderer:
            groups.append(("renderer", list(self.renderer.parameters()), lr.generator))
        if self.bkg_feats is not None and self.args.geoms.background.learnable:
            groups.append(("bkg_feats", [self.bkg_feats], lr.bkg_feats))
        return groups

    def init_optimizers(self, total_steps):
        lr = self.args.training.lr
        print("LR factor: ", lr.lr_factor)
        self.optimizers, self.schedulers = {}, {}
        for name, params, opt in self._groups():
            wd = 0 if name == "points" else opt.weight_decay
            # one launch per optimizer instead of five (in-box A/B: 0.06-0.17 ms per step, same losses to the last digit)
            fused = os.environ.get("PAPR_FUSED_ADAM", "1") == "1" and all(p.is_cuda for p in params)
            self.optimizers[name] = torch.optim.Adam(params, lr=opt.base_lr * lr.lr_factor, weight_decay=wd, **({"fused": True} if fused else {}))
            self.schedulers[name] = create_learning_rate_fn(self.optimizers[name], self.args.training.steps, opt)
        for name in self.args.training.fix_keys:
            if name in self.optimizers:
                print("Fixing {}".format(name))
                self.optimizers.pop(name)
                self.schedulers.pop(name)
        if total_steps > 0:                          # closed form instead of the reference's replay loop (schedule.fast_forward)
            opts = {name: opt for name, _, opt in self._groups()}
            for name, sched in self.schedulers.items():
                fast_forward(sched, opts[name], self.args.training.steps, total_steps, lr.lr_factor)

    def clear_optimizer(self):
        self.optimizers.clear()
        del self.optimizers

    def clear_scheduler(self):
        self.schedulers.clear()
        del self.schedulers

    def clear_grad(self):
        for opt in self.optimizers.values():
            if opt is not None:
                opt.zero_grad()

    def step(self, step=-1):
        if pdist.active():
            pdist.average_gradients([p for o in self.optimizers.values() if o is not None
                                     for g in o.param_groups for p in g["params"]])
        opts = [opt for opt in self.optimizers.values() if opt is not None]
        self.proximity_attn._kw_cache = None        # (papr_adam_step writes the parameters through raw pointers: no version bump)
        # one launch for all of them (papr_adam_step) where the reference's loop is `opt.step()` anyway: GradScaler off, plain Adam on
        # device parameters; PAPR_OWN_ADAM=0: torch's optimizers
        if _OWN_ADAM and not self.scaler.is_enabled() and own_adam.supported(opts):
            own_adam.step(opts)
        elif _OWN_ADAM and _OWN_ADAM_AMP and self.scaler.is_enabled() and own_adam.supported(opts) and own_adam.scaler_supported(self.scaler):
            # `use_amp: true`: the same launches under the GradScaler (torch's route: one inf-check launch and one fused-Adam launch per optimizer,
            # ~0.5 ms per step; here: one check pass, one step pass, the scaler's bookkeeping kept in step)
            own_adam.step_scaled(opts, self.scaler)
        else:
            for opt in opts:
                self.scaler.step(opt)
        for sched in self.schedulers.values():
            if sched is not None:
                sched.step()
        self.attn_lr = self._current_lr("attn")
        self.pts_lr = self._current_lr("points")

    def _current_lr(self, name):
        if name not in self.optimizers:
            return 0
        sched = self.schedulers[name]
        return sched.get_last_lr()[0] if sched is not None else self.optimizers[name].param_groups[0]["lr"]

    # --------------------------------------------------------------------------------- prune / add
    def _replace_points(self, points, influ, feats):
        dev = self.points.device
        self.points = nn.Parameter(points.to(dev), requires_grad=self.points.requires_grad)
        self.points_influ_scores = nn.Parameter(influ.to(dev), requires_grad=self.points_influ_scores.requires_grad)
        if self.use_pc_feats:
            self.pc_feats = nn.Parameter(feats.to(dev), requires_grad=self.pc_feats.requires_grad)

    def _sync_points(self):
        if pdist.active():
            tensors = [self.points.data, self.points_influ_scores.data] + ([self.pc_feats.data] if self.use_pc_feats else [])
            out = pdist.broadcast_point_cloud(tensors)
            self._replace_points(out[0], out[1], out[2] if self.use_pc_feats else None)

    def prune_points(self, thresh):
        if self.points_influ_scores is None:
            return 0
        s = self.points_influ_scores[:, 0]
        keep = (s > thresh) if self.args.training.prune_type == "<" else (s < thresh)
        n_drop = torch.sum(keep == 0)
        print("@@@@@@@@@  pruned {}/{}".format(n_drop, keep.shape[0]))
        self._replace_points(self.points.data[keep, :], self.points_influ_scores.data[keep, :],
                             self.pc_feats.data[keep, :] if self.use_pc_feats else None)
        self._sync_points()
        print("@@@@@@@@@ New points: ", self.points.shape)
        return n_drop

    def add_points(self, add_num):
        cur = self.points.shape[0]
        if "max_points" in self.args and self.args.max_points > 0 and (cur + add_num) >= self.args.max_points:
            add_num = self.args.max_points - cur
            if add_num <= 0:
                return 0
        g = self.args.geoms.points
        kw = dict(add_num=add_num, k=g.add_k, comb_type=g.add_type, sample_k=g.add_sample_k, sample_type=g.add_sample_type)
        if self.points.is_cuda:     # the cloud stays on the device (papr_points_knn); the numpy draws are the host's
            pts = self.points.detach()
            new_pts, n_new, new_influ, new_feats = grow_points_device(pts, self.points_influ_scores, feats=self.pc_feats if self.use_pc_feats else None, **kw)
        else:                       # a model on the CPU: the reference's own host procedure
            pts = self.points.detach().cpu()
            new_pts, n_new, new_influ, new_feats = grow_points(pts, self.points_influ_scores.detach().cpu(),
                                                               feats=self.pc_feats.detach().cpu() if self.use_pc_feats else None, **kw)
        print("@@@@@@@@@  added {} points".format(n_new))
        if n_new > 0:
            dev = self.points.device
            self._replace_points(torch.cat([pts, new_pts.to(dev)], dim=0),
                                 torch.cat([self.points_influ_scores.data, new_influ.to(dev)], dim=0),
                                 torch.cat([self.pc_feats.data, new_feats.to(dev)], dim=0) if self.use_pc_feats else None)
            self._sync_points()
            print("@@@@@@@@@ New points: ", self.points.shape)
        return n_new

    # ------------------------------------------------------------------------------- render path
    def _render(self, rays_o, rays_d):
        """rays_o (N,3), rays_d (N,H,W,3) -> fused (R,C), attn (R,k+1), idx (R,k) int32, sel (R*k,3)."""
        if not self.points.is_cuda:
            raise RuntimeError("papr_amd: the render path runs only on a ROCm device (HIP kernels); there is no CPU "
                               "fallback -- move the model and rays to 'cuda'")
        N, H, W, _ = rays_d.shape
        ro = rays_o.reshape(N, 3).contiguous().float()
        rd = rays_d.reshape(-1, 3).contiguous().float()
        idx = self.plan.select(self.points.detach(), ro, rd, H * W)
        weights = self.proximity_attn.kernel_weights(self.plan)
        feats = self.pc_feats if self.use_pc_feats else None
        fused, attn, sel = render_rays(self.plan, ro, rd, H * W, idx, self.points, feats, self.points_influ_scores, weights)
        return fused, attn, idx, sel

    def evaluate(self, rays_o, rays_d, c2w, step=-1, shading_code=None):
        N, H, W, _ = rays_d.shape
        fused, attn, idx, sel = self._render(rays_o, rays_d)
        k = idx.shape[-1]
        self.select_k_ind = idx.reshape(N, H, W, k).long()
        self.selected_points = sel.reshape(N, H, W, k, 3)
        return fused.reshape(N, H, W, 1, -1), attn.reshape(N, H, W, k + 1, 1)

    def forward(self, rays_o, rays_d, c2w, step=-1, shading_code=None):
        N, H, W, _ = rays_d.shape
        fused, attn, idx, sel = self._render(rays_o, rays_d)
        k = idx.shape[-1]
        self.selected_points = sel.reshape(N, H, W, k, 3)
        fmap = fused.reshape(N, H, W, -1)
        if self.args.models.use_renderer:
            fg = self.renderer(fmap.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).float()
        else:
            fg = fmap
        if _OWN_COMPOSITE and fg.is_cuda and fg.dtype == torch.float32 and fg.shape[-1] <= 8 and fg.shape[-1] == self.bkg_feats.numel():
            from .ops import _CompositeFn                    # (one launch forward, two backward: papr_composite_fwd / _bwd)
            return _CompositeFn.apply(fg.contiguous(), attn.reshape(-1, k + 1), self.bkg_feats, bool(self.args.models.normalize_topk_attn))
        bkg_attn = attn.reshape(N, H, W, k + 1)[..., k:]
        bkg = self.bkg_feats.reshape(1, 1, 1, -1)
        if self.args.models.normalize_topk_attn:
            rgb = fg * (1 - bkg_attn) + bkg * bkg_attn
        else:
            rgb = fg + bkg * bkg_attn
        return rgb

    # --------------------------------------------------------------------------------- checkpoints
    def save(self, step, save_dir):
        torch.save({str(step): self.state_dict()}, os.path.join(save_dir, "model.pth"))
        torch.save({n: (o.state_dict() if o is not None else None) for n, o in self.optimizers.items()},
                   os.path.join(save_dir, "optimizers.pth"))
        torch.save({n: (s.state_dict() if s is not None else None) for n, s in self.schedulers.items()},
                   os.path.join(save_dir, "schedulers.pth"))
        torch.save(self.scaler.state_dict(), os.path.join(save_dir, "scaler.pth"))

    def load(self, load_dir, load_optimizer=False):
        if load_optimizer:
            osd = torch.load(os.path.join(load_dir, "optimizers.pth"))
            for n, o in self.optimizers.items():
                if o is not None:
                    o.load_state_dict(osd[n])
            ssd = torch.load(os.path.join(load_dir, "schedulers.pth"))
            for n, s in self.schedulers.items():
                if s is not None:
                    s.load_state_dict(ssd[n])
        scaler_path = os.path.join(load_dir, "scaler.pth")
        if os.path.exists(scaler_path):
            sd = torch.load(scaler_path)
            if sd:
                self.scaler.load_state_dict(sd)
        for step, sd in torch.load(os.path.join(load_dir, "model.pth"), map_location="cpu").items():
            self.load_my_state_dict(sd)
            return int(step)

    def load_my_state_dict(self, state_dict, exclude_keys=[]):
        own = self.state_dict()
        per_point = ("points", "points_influ_scores", "pc_feats")
        for name, value in state_dict.items():
            if any(x in name for x in exclude_keys):
                print("exclude", name)
                continue
            if name in per_point:
                continue
            value = value.data if isinstance(value, nn.Parameter) else value
            try:
                own[name].copy_(value)
            except Exception:
                print("Can't load", name)
        dev = self.points.device
        self._replace_points(state_dict["points"].data.to(dev), state_dict["points_influ_scores"].data.to(dev),
                             state_dict["pc_feats"].data.to(dev) if self.use_pc_feats else None)
