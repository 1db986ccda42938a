"""Kernel-level parity: every C-ABI entry point of libpapr_hip.so against the CPU oracle.

All tests here need an MI355X (`-m gpu`).  Tolerances are fp32 round-off scaled to the magnitudes
involved; index work (kNN) is compared exactly against the reference's golden index sets.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import golden, case_cfg
from formula import synth_rays, uniform_points
from oracle import papr_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def dev():
    return torch.device("cuda:0")


def test_library_loads_on_device():
    from papr_amd import hip
    assert hip.lib().papr_abi_version() == hip.EXPECTED_ABI


# ------------------------------------------------------------------------------------------- K1
def _knn(points, ro, rd, k, eps=1e-6):
    from papr_amd import ops
    N, H, W, _ = rd.shape
    p_d, ro_d, rd_d = points.to(dev()), ro.to(dev()), rd.reshape(-1, 3).contiguous().to(dev())
    idx, dist = ops.ray_knn(p_d, ro_d, rd_d, H * W, k, eps, want_dist=True)
    torch.cuda.synchronize()
    return idx.cpu(), dist.cpu()


def _check_knn(points, ro, rd, k, ref_sorted_sets):
    idx, dist = _knn(points, ro, rd, k)
    got = np.sort(idx.numpy(), -1)
    ref = ref_sorted_sets.reshape(-1, k)
    mism = (got != ref).any(-1)
    # distances as the oracle computes them for the returned indices: must be ascending and equal
    feat = O.ray_point_distance(points, ro, rd, 1e-6).reshape(-1, points.shape[0])
    d_or = torch.gather(feat, 1, idx.long())
    assert torch.all(d_or[:, 1:] >= d_or[:, :-1] - 1e-6), "kernel output not sorted by distance"
    np.testing.assert_allclose(dist.numpy(), d_or.numpy(), rtol=2e-6, atol=1e-6)
    if mism.any():
        # a different set is only acceptable at an exact distance tie with the reference's k-th neighbour
        kth_ref = torch.gather(feat, 1, T(ref).long()).max(-1).values
        kth_got = d_or.max(-1).values
        assert torch.equal(kth_ref[T(mism)], kth_got[T(mism)]), "kNN sets differ on %d rays (not ties)" % mism.sum()
    return int(mism.sum())


def test_knn_lattice_cloud_matches_reference_sets():
    g = golden("g34_knn_geometry.npz")
    pts = T(g["a_points"])
    ro, rd, _ = synth_rays(1, 16, 16, seed=0)
    assert _check_knn(pts, ro, rd, 20, g["a_idx"]) == 0


def test_knn_10k_uniform_cloud_matches_reference_sets():
    g = golden("g34_knn_geometry.npz")
    pts = uniform_points(10000, 12.0, seed=5)
    ro, rd, _ = synth_rays(1, 32, 32, seed=3)
    assert _check_knn(pts, ro, rd, 20, g["b_idx"]) == 0


def test_knn_two_origins_and_unnormalised_directions():
    g = golden("g34_knn_geometry.npz")
    pts = T(g["a_points"])
    ro, rd, _ = synth_rays(2, 8, 8, seed=7)
    assert _check_knn(pts, ro, rd, 20, g["c_idx"]) == 0
    ro, rd, _ = synth_rays(1, 8, 8, seed=9)
    assert _check_knn(pts, ro, rd * 1.7, 20, g["d_idx"]) == 0


@pytest.mark.parametrize("P,k,hw", [(64, 64, 5), (65, 1, 3), (257, 7, 9), (30000, 20, 40), (1000, 33, 7)])
def test_knn_ragged_sizes_against_oracle(P, k, hw):
    pts = uniform_points(P, 12.0, seed=P + k)
    ro, rd, _ = synth_rays(1, hw, hw, seed=P)
    ref, _ = O.knn_select(pts, ro, rd, k, 1e-6)
    _check_knn(pts, ro, rd, k, np.sort(ref.numpy(), -1))


def test_knn_duplicate_points_exact_ties():
    pts = uniform_points(500, 12.0, seed=1)
    pts = torch.cat([pts, pts[:100]])            # exact duplicates -> exact ties
    ro, rd, _ = synth_rays(1, 6, 6, seed=2)
    idx, dist = _knn(pts, ro, rd, 10)
    assert torch.all(dist[:, 1:] >= dist[:, :-1])
    feat = O.ray_point_distance(pts, ro, rd, 1e-6).reshape(-1, pts.shape[0])
    kth = feat.topk(10, largest=False).values.max(-1).values
    np.testing.assert_allclose(dist.max(-1).values.numpy(), kth.numpy(), rtol=2e-6)
    # the output is ascending in (distance, index)
    same = dist[:, 1:] == dist[:, :-1]
    assert torch.all(idx[:, 1:][same] > idx[:, :-1][same])
    # the set is the k smallest by (distance, index) in this form too: of a duplicated pair never only the higher index
    for r in range(idx.shape[0]):
        members = set(idx[r].tolist())
        for j in members:
            if j >= 500:
                assert j - 500 in members, (r, j)


def test_knn_block_path_total_order_and_determinism():
    """P >= 2,048 runs the spatial form (binned cloud, bounding spheres): exact duplicates across blocks, a run-to-run identical
    answer although the binning places points with atomics, and -- a cloud of ONE repeated point -- the k smallest indices, because
    the set is the k smallest by (distance, index)."""
    pts = uniform_points(3000, 12.0, seed=11)
    pts = torch.cat([pts, pts[:700]])            # exact duplicates, far apart in index
    ro, rd, _ = synth_rays(1, 9, 9, seed=4)
    idx, dist = _knn(pts, ro, rd, 10)
    idx2, dist2 = _knn(pts, ro, rd, 10)
    assert torch.equal(idx, idx2) and torch.equal(dist, dist2)
    feat = O.ray_point_distance(pts, ro, rd, 1e-6).reshape(-1, pts.shape[0])
    kth = feat.topk(10, largest=False).values.max(-1).values
    np.testing.assert_allclose(dist.max(-1).values.numpy(), kth.numpy(), rtol=2e-6)
    same = dist[:, 1:] == dist[:, :-1]
    assert torch.all(idx[:, 1:][same] > idx[:, :-1][same])
    # a duplicated pair is two members of the same distance: both or the lower index, never only the higher one
    lowdup = idx[(idx >= 3000)] - 3000
    for r in range(idx.shape[0]):
        members = set(idx[r].tolist())
        for j in members:
            if j >= 3000:
                assert j - 3000 in members, (r, j)
    one = torch.tensor([[0.3, -0.2, 0.1]]).repeat(2500, 1)
    idx, dist = _knn(one, ro, rd, 7)
    assert torch.equal(idx, torch.arange(7, dtype=idx.dtype).expand_as(idx))


@pytest.mark.parametrize("P,k,hw", [(300, 100, 6), (130, 128, 4), (5000, 65, 8), (1000, 200, 5), (257, 256, 3)])
def test_knn_more_than_64_neighbours_against_oracle(P, k, hw):
    """64 < k <= 256 (ABI 25; the reference's topk takes any k < P, models/model.py:281): the wide form -- the set across 2 or 4 registers per lane --
    gives the oracle's sets, ascending in (distance, index)."""
    pts = uniform_points(P, 12.0, seed=P + k)
    ro, rd, _ = synth_rays(1, hw, hw, seed=P)
    ref, _ = O.knn_select(pts, ro, rd, k, 1e-6)
    _check_knn(pts, ro, rd, k, np.sort(ref.numpy(), -1))


def test_knn_wide_form_exact_ties():
    pts = uniform_points(300, 12.0, seed=3)
    pts = torch.cat([pts, pts[:150]])            # exact duplicates -> exact ties, also at the k-th distance
    ro, rd, _ = synth_rays(1, 5, 5, seed=2)
    k = 90
    idx, dist = _knn(pts, ro, rd, k)
    assert torch.all(dist[:, 1:] >= dist[:, :-1])
    feat = O.ray_point_distance(pts, ro, rd, 1e-6).reshape(-1, pts.shape[0])
    kth = feat.topk(k, largest=False).values.max(-1).values
    np.testing.assert_allclose(dist.max(-1).values.numpy(), kth.numpy(), rtol=2e-6)
    same = dist[:, 1:] == dist[:, :-1]
    assert torch.all(idx[:, 1:][same] > idx[:, :-1][same])
    for r in range(idx.shape[0]):
        members = set(idx[r].tolist())
        assert len(members) == k
        for j in members:
            if j >= 300:
                assert j - 300 in members, (r, j)


@pytest.mark.parametrize("P,k", [(135, 20), (3000, 20), (3000, 100)])
def test_knn_of_a_cloud_of_nans_returns_indices_inside_the_cloud(P, k):
    """A training run that has diverged hands the neighbour search NaN coordinates: no candidate ever enters a ray's set.  Every output slot must still hold an
    index inside the cloud (torch.topk of NaN distances returns some) -- the slots' initial -1 sent the gathers of features_fwd out of bounds: the memory access
    fault at the end of round 6's long lego run.  All three forms: small cloud, binned cloud, more than 64 neighbours."""
    pts = torch.full((P, 3), float("nan"))
    ro, rd, _ = synth_rays(1, 16, 16, seed=3)
    for cloud in (pts, torch.cat([uniform_points(k // 2, 12.0, seed=1), pts[k // 2:]])):      # all NaN; fewer finite points than k
        idx, _ = _knn(cloud, ro, rd, k)
        assert int(idx.min()) >= 0 and int(idx.max()) < P


def test_knn_rejects_bad_k():
    from papr_amd import ops
    pts = uniform_points(300, 1.0, seed=0).to(dev())
    ro, rd, _ = synth_rays(1, 2, 2)
    with pytest.raises(RuntimeError):
        ops.ray_knn(pts, ro.to(dev()), rd.reshape(-1, 3).to(dev()), 4, 257, 1e-6)


# ------------------------------------------------------------------------------------------- K2
def _plan(tag):
    from papr_amd.ops import RenderPath
    cfg = case_cfg(tag)
    return cfg, RenderPath(cfg)


@pytest.mark.parametrize("tag", ["chair1k", "tiny_norender"])
def test_features_forward_match_oracle(tag):
    from papr_amd import hip
    g = golden("g567_%s.npz" % tag)
    cfg, plan = _plan(tag)
    from conftest import case_rays
    ro, rd, _ = case_rays(tag)
    idx = T(g["idx_raw"]).long()
    P = g["points"].shape[0]
    st = {"points": T(g["points"]), "pc_feats": torch.randn(P, 64, generator=torch.Generator().manual_seed(3))}
    key_o, qry_o, val_o, sel_o, _, _ = O.build_inputs(st, cfg, ro, rd, idx)
    R, k = idx.reshape(-1, idx.shape[-1]).shape
    fd = plan.feature_desc(k)
    d = dev()
    key = torch.empty((R * k, plan.key.ld_in), device=d)
    qry = torch.empty((R, plan.qry.ld_in), device=d)
    val = torch.empty((R * k, plan.val.ld_in), device=d)
    sel = torch.empty((R * k, 3), device=d)
    # keep every device tensor alive in a local: a temporary would be freed (and reused) before the launch
    pts_d, f_d, ro_d, rd_d = st["points"].to(d), st["pc_feats"].to(d), ro.to(d), rd.reshape(-1, 3).contiguous().to(d)
    idx_d = idx.reshape(R, k).int().to(d)
    # (with the statistics of the LayerNorm core in front of the key MLP where the key carries no point features: ABI 24)
    want_stats = not fd.key_has_feats
    kstats = torch.empty((R * k, 2), device=d) if want_stats else None
    kmean = torch.empty((R * k,), device=d) if want_stats else None
    hip.check(hip.lib().papr_build_features_fwd(C.byref(fd), hip.ptr(pts_d), hip.ptr(f_d), hip.ptr(ro_d), hip.ptr(rd_d), R,
                                                rd.shape[1] * rd.shape[2], hip.ptr(idx_d), hip.ptr(key), hip.ptr(qry), hip.ptr(val),
                                                hip.ptr(sel), hip.ptr(kstats), hip.ptr(kmean), float(cfg["eps"]), hip.stream_ptr()), "features_fwd")
    torch.cuda.synchronize()
    kw, qw, vw = plan.key_w, plan.qry_w, plan.val_w
    if want_stats:      # mean, unbiased std and 1 / (std + eps) of every key row, against torch on the rows the kernel wrote
        rows = key.cpu()[:, :kw].double()
        np.testing.assert_allclose(kmean.cpu().numpy(), rows.mean(1).numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(kstats.cpu()[:, 1].numpy(), rows.std(1).numpy(), rtol=2e-6, atol=0)
        np.testing.assert_allclose(kstats.cpu()[:, 0].numpy(), (1.0 / (rows.std(1) + cfg["eps"])).numpy(), rtol=2e-6, atol=0)
    assert torch.equal(sel.cpu(), sel_o.reshape(-1, 3))
    # x, s, u reproduce the reference bit for bit; sin/cos differ by libm (<= 2 ulp of 1.0)
    np.testing.assert_allclose(key.cpu()[:, :kw].numpy(), key_o.reshape(-1, kw).numpy(), rtol=0, atol=5e-7)
    np.testing.assert_allclose(qry.cpu()[:, :qw].numpy(), qry_o.reshape(-1, qw).numpy(), rtol=0, atol=5e-7)
    np.testing.assert_allclose(val.cpu()[:, :vw].numpy(), val_o.reshape(-1, vw).numpy(), rtol=0, atol=5e-7)
    assert torch.all(key.cpu()[:, kw:] == 0) and torch.all(val.cpu()[:, vw:] == 0) and torch.all(qry.cpu()[:, qw:] == 0)
    per = 1 + 2 * cfg["models"]["attn"]["embed"]["k_L"][1]
    raw_cols = [3 * per + c * per for c in range(3)]     # the un-encoded s components inside the key row
    assert torch.equal(key.cpu()[:, raw_cols], key_o.reshape(-1, kw)[:, raw_cols]), "s = r t must be bit-exact"


def test_features_backward_matches_autograd():
    from papr_amd import hip
    tag = "chair1k"
    g = golden("g567_%s.npz" % tag)
    cfg, plan = _plan(tag)
    from conftest import case_rays
    ro, rd, _ = case_rays(tag)
    idx = T(g["idx_raw"]).long()
    P = g["points"].shape[0]
    gen = torch.Generator().manual_seed(5)
    st = {"points": T(g["points"]).clone().requires_grad_(True),
          "pc_feats": torch.randn(P, 64, generator=gen).requires_grad_(True)}
    key_o, _, val_o, _, _, _ = O.build_inputs(st, cfg, ro, rd, idx)
    gk = torch.randn(key_o.shape, generator=gen)
    gv = torch.randn(val_o.shape, generator=gen)
    (key_o * gk).sum().add((val_o * gv).sum()).backward()
    R, k = idx.reshape(-1, idx.shape[-1]).shape
    d = dev()
    fd = plan.feature_desc(k)
    gk_p = torch.zeros((R * k, plan.key.ld_in)); gk_p[:, :plan.key_w] = gk.reshape(R * k, -1)
    gv_p = torch.zeros((R * k, plan.val.ld_in)); gv_p[:, :plan.val_w] = gv.reshape(R * k, -1)
    d_pts = torch.zeros((P, 3), device=d)
    d_f = torch.zeros((P, 64), device=d)
    pts_d, ro_d, rd_d = st["points"].detach().to(d), ro.to(d), rd.reshape(-1, 3).contiguous().to(d)
    idx_d, gk_d, gv_d = idx.reshape(R, k).int().to(d), gk_p.to(d), gv_p.to(d)
    hip.check(hip.lib().papr_build_features_bwd(C.byref(fd), hip.ptr(pts_d), hip.ptr(ro_d), hip.ptr(rd_d), R, rd.shape[1] * rd.shape[2],
                                                hip.ptr(idx_d), hip.ptr(gk_d), hip.ptr(gv_d), hip.ptr(d_pts), hip.ptr(d_f),
                                                hip.stream_ptr()), "features_bwd")
    torch.cuda.synchronize()
    ref = st["points"].grad
    np.testing.assert_allclose(d_pts.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-4 * ref.abs().max().item())
    fref = st["pc_feats"].grad          # sums of up to a few hundred N(0,1) terms, atomics add in any order
    np.testing.assert_allclose(d_f.cpu().numpy(), fref.numpy(), rtol=0, atol=3e-6 * fref.abs().max().item())


def _segment_inputs(flat, P):
    """What ops._RenderFn hands papr_segment_reduce: pairs grouped by point (stable), group bounds (papr_group_pairs)."""
    from papr_amd import ops
    return ops.group_pairs(flat, P)


@pytest.mark.parametrize("P,M,hot", [(30000, 512000, 0), (10000, 512000, 40000), (257, 1283, 0), (1, 77, 0), (5, 1, 0), (70000, 6, 0), (9, 0, 0)])
def test_group_pairs_is_the_stable_sort_with_group_bounds(P, M, hot):
    """papr_group_pairs (pairs.hip) WITHOUT a promise about its input (run = 1: one entry per step) against torch.sort(stable) + bincount +
    cumsum on the CPU, bit for bit: the permutation (pair ids ascending inside a group), the sorted keys and the P + 1 bounds; full-size, one
    hot point, a single point, a single pair, more points than pairs, no pairs."""
    from papr_amd import ops
    gen = torch.Generator().manual_seed(3 * P + M)
    flat = torch.randint(0, P, (M,), generator=gen).int()
    if hot:
        flat[torch.randperm(M, generator=gen)[:hot]] = P // 2
    want_pts, want_order = torch.sort(flat, stable=True)
    want_seg = torch.zeros(P + 1, dtype=torch.int64)
    torch.cumsum(torch.bincount(flat, minlength=P), 0, out=want_seg[1:])
    order, sorted_pts, seg = ops.group_pairs(flat.to("cuda:0"), P)
    torch.cuda.synchronize()
    assert torch.equal(order.cpu(), want_order)
    assert torch.equal(sorted_pts.cpu(), want_pts)
    assert torch.equal(seg.cpu(), want_seg)


@pytest.mark.parametrize("P,R,k", [(10000, 25600, 20), (30000, 25600, 20), (50000, 3000, 63), (64, 500, 64), (300, 41, 300), (25, 1000, 7)])
def test_group_pairs_with_runs_of_distinct_points_is_the_stable_sort(P, R, k):
    """The product's call: `run = k`, every ray's k selected points are distinct (a kNN result; or all P points of a small cloud, k = P > 64).
    One returning LDS add per run places its entries; the result is torch.sort(stable) bit for bit -- incl. more points than one pass of LDS
    counters holds (P = 50,000), runs longer than a wave (k = 300), and a last workgroup with a short chunk."""
    from papr_amd import ops
    gen = torch.Generator().manual_seed(P + R + k)
    # k distinct points per ray, neighbouring rays share most of theirs (like a render): a window of the cloud that moves with the ray
    base = (torch.arange(R) * 3) % max(P - 2 * k, 1)
    flat = torch.stack([base[r] + torch.randperm(min(2 * k, P), generator=gen)[:k] for r in range(R)]).reshape(-1).int() % P if k < P else \
        torch.stack([torch.randperm(P, generator=gen) for _ in range(R)]).reshape(-1).int()
    assert all(len(set(row.tolist())) == k for row in flat.view(R, k)[:50])
    M = R * k
    want_pts, want_order = torch.sort(flat, stable=True)
    want_seg = torch.zeros(P + 1, dtype=torch.int64)
    torch.cumsum(torch.bincount(flat, minlength=P), 0, out=want_seg[1:])
    order, sorted_pts, seg = ops.group_pairs(flat.to("cuda:0"), P, run=k)
    torch.cuda.synchronize()
    assert torch.equal(order.cpu(), want_order)
    assert torch.equal(sorted_pts.cpu(), want_pts)
    assert torch.equal(seg.cpu(), want_seg)


@pytest.mark.parametrize("P,M,hot", [(30000, 200000, 0), (1000, 50000, 3000), (257, 1283, 0), (5, 128, 0), (3, 700, 650)])
def test_segment_reduce_matches_index_put_accumulate(P, M, hot):
    """The product path of the three gather backwards (ops.py: papr_segment_reduce) against torch CPU
    index_put_(accumulate=True) -- the reference's semantics (models/model.py:330, 431-435, 509).  Cases: the end-of-training
    cloud with many never-selected points (rows stay zero), one point owning thousands of pairs (its group spans dozens of
    128-entry chunks: atomics), sizes that are not multiples of the chunk, fewer points than lanes."""
    from papr_amd import hip
    gen = torch.Generator().manual_seed(P + M)
    live = torch.randperm(P, generator=gen)[: max(1, (2 * P) // 3)]          # a third of the points is never selected
    flat = live[torch.randint(0, live.numel(), (M,), generator=gen)].int()
    if hot:
        flat[torch.randperm(M, generator=gen)[:hot]] = int(live[0])
    pair_pts = torch.randn(M, 4, generator=gen)
    pair_influ = torch.randn(M, generator=gen)
    ld, col0, nc = 144, 78, 64
    rows = torch.randn(M, ld, generator=gen)
    want_p = torch.zeros(P, 3, dtype=torch.float64).index_put_((flat.long(),), pair_pts[:, :3].double(), accumulate=True)
    want_i = torch.zeros(P, dtype=torch.float64).index_put_((flat.long(),), pair_influ.double(), accumulate=True)
    want_f = torch.zeros(P, nc, dtype=torch.float64).index_put_((flat.long(),), rows[:, col0:col0 + nc].double(), accumulate=True)
    d = dev()
    flat_d = flat.to(d)
    order, sorted_pts, seg = _segment_inputs(flat_d, P)
    d_p, d_i, d_f = torch.zeros(P, 3, device=d), torch.zeros(P, 1, device=d), torch.zeros(P, nc, device=d)
    a = [hip.ptr(order), hip.ptr(sorted_pts), hip.ptr(seg), M, P]
    pp_d, pi_d, rows_d = pair_pts.to(d), pair_influ.to(d), rows.to(d)          # (kept alive: hip.ptr holds no reference)
    ws = torch.empty(hip.lib().papr_segment_reduce_workspace_bytes(M) // 4, device=d)
    hip.check(hip.lib().papr_segment_reduce(*a, hip.ptr(pp_d), hip.ptr(pi_d), hip.ptr(rows_d), ld, col0, nc,
                                            hip.ptr(d_p), hip.ptr(d_i), hip.ptr(d_f), 0, hip.ptr(ws), hip.stream_ptr()), "papr_segment_reduce")
    torch.cuda.synchronize()
    # no atomics: a second run gives the same bits (a popular point's group spans dozens of chunks)
    e_p, e_i, e_f = torch.zeros(P, 3, device=d), torch.zeros(P, 1, device=d), torch.zeros(P, nc, device=d)
    hip.check(hip.lib().papr_segment_reduce(*a, hip.ptr(pp_d), hip.ptr(pi_d), hip.ptr(rows_d), ld, col0, nc,
                                            hip.ptr(e_p), hip.ptr(e_i), hip.ptr(e_f), 0, hip.ptr(ws), hip.stream_ptr()), "papr_segment_reduce")
    torch.cuda.synchronize()
    assert torch.equal(e_p, d_p) and torch.equal(e_i, d_i) and torch.equal(e_f, d_f)
    big = lambda w: 4e-6 * w.abs().max().item() + 1e-7          # fp32 sums of up to `hot` terms in another order
    np.testing.assert_allclose(d_p.cpu().double().numpy(), want_p.numpy(), rtol=0, atol=big(want_p))
    np.testing.assert_allclose(d_i.cpu()[:, 0].double().numpy(), want_i.numpy(), rtol=0, atol=big(want_i))
    np.testing.assert_allclose(d_f.cpu().double().numpy(), want_f.numpy(), rtol=0, atol=big(want_f))
    never = torch.ones(P, dtype=torch.bool); never[flat.long()] = False
    assert torch.all(d_f.cpu()[never] == 0) and torch.all(d_p.cpu()[never] == 0)
    # second pass onto the same outputs (use_ink + use_inv: features feed the key and the value branch): sums ADD
    rows2 = torch.randn(M, 120, generator=gen)
    want_f2 = want_f + torch.zeros(P, nc, dtype=torch.float64).index_put_((flat.long(),), rows2[:, 56:56 + nc].double(), accumulate=True)
    rows2_d = rows2.to(d)
    hip.check(hip.lib().papr_segment_reduce(*a, None, None, hip.ptr(rows2_d), 120, 56, nc, None, None, hip.ptr(d_f), 1,
                                            hip.ptr(ws), hip.stream_ptr()), "papr_segment_reduce")
    torch.cuda.synchronize()
    np.testing.assert_allclose(d_f.cpu().double().numpy(), want_f2.numpy(), rtol=0, atol=big(want_f2))


def test_features_backward_pairs_plus_segment_reduce_is_the_product_path():
    """papr_build_features_bwd_pairs + papr_segment_reduce (what ops._RenderFn.backward calls) against autograd through the
    oracle's feature construction -- same case as the atomic variant above."""
    from papr_amd import hip
    tag = "chair1k"
    g = golden("g567_%s.npz" % tag)
    cfg, plan = _plan(tag)
    from conftest import case_rays
    ro, rd, _ = case_rays(tag)
    idx = T(g["idx_raw"]).long()
    P = g["points"].shape[0]
    gen = torch.Generator().manual_seed(5)
    st = {"points": T(g["points"]).clone().requires_grad_(True), "pc_feats": torch.randn(P, 64, generator=gen).requires_grad_(True)}
    key_o, _, val_o, _, _, _ = O.build_inputs(st, cfg, ro, rd, idx)
    gk, gv = torch.randn(key_o.shape, generator=gen), torch.randn(val_o.shape, generator=gen)
    (key_o * gk).sum().add((val_o * gv).sum()).backward()
    R, k = idx.reshape(-1, idx.shape[-1]).shape
    M = R * k
    d = dev()
    fd = plan.feature_desc(k)
    gk_p = torch.zeros((M, plan.key.ld_in)); gk_p[:, :plan.key_w] = gk.reshape(M, -1)
    gv_p = torch.zeros((M, plan.val.ld_in)); gv_p[:, :plan.val_w] = gv.reshape(M, -1)
    pts_d, ro_d, rd_d = st["points"].detach().to(d), ro.to(d), rd.reshape(-1, 3).contiguous().to(d)
    idx_d, gk_d, gv_d = idx.reshape(R, k).int().to(d), gk_p.to(d), gv_p.to(d)
    pair_pts = torch.empty((M, 4), device=d)
    hip.check(hip.lib().papr_build_features_bwd_pairs(C.byref(fd), hip.ptr(pts_d), hip.ptr(ro_d), hip.ptr(rd_d), R, rd.shape[1] * rd.shape[2],
                                                      hip.ptr(idx_d), hip.ptr(gk_d), hip.ptr(gv_d), hip.ptr(pair_pts), None, None, hip.stream_ptr()), "bwd_pairs")
    order, sorted_pts, seg = _segment_inputs(idx_d.view(-1), P)
    d_pts, d_f = torch.zeros((P, 3), device=d), torch.zeros((P, 64), device=d)
    ws = torch.empty(hip.lib().papr_segment_reduce_workspace_bytes(M) // 4, device=d)
    hip.check(hip.lib().papr_segment_reduce(hip.ptr(order), hip.ptr(sorted_pts), hip.ptr(seg), M, P, hip.ptr(pair_pts), None, hip.ptr(gv_d),
                                            gv_d.shape[1], plan.val_w - 64, 64, hip.ptr(d_pts), None, hip.ptr(d_f), 0, hip.ptr(ws), hip.stream_ptr()), "segment_reduce")
    torch.cuda.synchronize()
    ref, fref = st["points"].grad, st["pc_feats"].grad
    np.testing.assert_allclose(d_pts.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-4 * ref.abs().max().item())
    np.testing.assert_allclose(d_f.cpu().numpy(), fref.numpy(), rtol=0, atol=3e-6 * fref.abs().max().item())


def test_features_backward_with_the_key_layernorm_core_inside():
    """key_mean / key_stats of papr_build_features_bwd_pairs (ABI 25): d_key is the gradient w.r.t. the STANDARDISED key rows and the LayerNorm core's
    backward pass rides in the kernel -- against autograd through the oracle's feature construction + custom_layernorm (models/attn.py:39-42), and
    against the two-kernel form it replaces (papr_rownorm_bwd on the gradient rows, then the plain kernel)."""
    from papr_amd import hip, ops
    tag = "chair1k"
    g = golden("g567_%s.npz" % tag)
    cfg, plan = _plan(tag)
    from conftest import case_rays
    ro, rd, _ = case_rays(tag)
    idx = T(g["idx_raw"]).long()
    P = g["points"].shape[0]
    gen = torch.Generator().manual_seed(6)
    st = {"points": T(g["points"]).clone().requires_grad_(True), "pc_feats": torch.randn(P, 64, generator=gen)}
    key_o, _, val_o, _, _, _ = O.build_inputs(st, cfg, ro, rd, idx)
    w = plan.key_w
    key_n = O.custom_layernorm(key_o, torch.ones(w), torch.zeros(w), plan.eps)
    gk, gv = torch.randn(key_o.shape, generator=gen), torch.randn(val_o.shape, generator=gen)
    (key_n * gk).sum().add((val_o * gv).sum()).backward()
    R, k = idx.reshape(-1, idx.shape[-1]).shape
    M = R * k
    d = dev()
    fd = plan.feature_desc(k)
    pts_d, ro_d, rd_d = st["points"].detach().to(d), ro.to(d), rd.reshape(-1, 3).contiguous().to(d)
    idx_d = idx.reshape(R, k).int().to(d)
    # forward on the device: raw key rows + their statistics, as the render path gets them
    key_in = torch.empty((M, plan.key.ld_in), device=d); qry_in = torch.empty((R, plan.qry.ld_in), device=d); val_in = torch.empty((M, plan.val.ld_in), device=d)
    sel = torch.empty((M, 3), device=d); kstats = torch.empty((M, 2), device=d); kmean = torch.empty((M,), device=d)
    hip.check(hip.lib().papr_build_features_fwd(C.byref(fd), hip.ptr(pts_d), hip.ptr(st["pc_feats"].to(d)), hip.ptr(ro_d), hip.ptr(rd_d), R, rd.shape[1] * rd.shape[2],
                                                hip.ptr(idx_d), hip.ptr(key_in), hip.ptr(qry_in), hip.ptr(val_in), hip.ptr(sel), hip.ptr(kstats), hip.ptr(kmean),
                                                plan.eps, hip.stream_ptr()), "features_fwd")
    gk_p = torch.zeros((M, plan.key.ld_in)); gk_p[:, :w] = gk.reshape(M, -1)
    gv_p = torch.zeros((M, plan.val.ld_in)); gv_p[:, :plan.val_w] = gv.reshape(M, -1)
    gk_d, gv_d = gk_p.to(d), gv_p.to(d)
    pair_a, pair_b = torch.empty((M, 4), device=d), torch.empty((M, 4), device=d)
    hip.check(hip.lib().papr_build_features_bwd_pairs(C.byref(fd), hip.ptr(pts_d), hip.ptr(ro_d), hip.ptr(rd_d), R, rd.shape[1] * rd.shape[2],
                                                      hip.ptr(idx_d), hip.ptr(gk_d), hip.ptr(gv_d), hip.ptr(pair_a), hip.ptr(kmean), hip.ptr(kstats), hip.stream_ptr()), "bwd_pairs")
    # the two-kernel form: standardise the rows as the fused run's staging does, papr_rownorm_bwd, then the plain kernel
    y = ((key_in[:, :w] - kmean[:, None]) * kstats[:, :1])
    yp = torch.zeros_like(key_in); yp[:, :w] = y
    gk2 = gk_d.clone()
    ops.rownorm_bwd_(gk2, yp, kstats, w, plan.eps)
    hip.check(hip.lib().papr_build_features_bwd_pairs(C.byref(fd), hip.ptr(pts_d), hip.ptr(ro_d), hip.ptr(rd_d), R, rd.shape[1] * rd.shape[2],
                                                      hip.ptr(idx_d), hip.ptr(gk2), hip.ptr(gv_d), hip.ptr(pair_b), None, None, hip.stream_ptr()), "bwd_pairs")
    order, sorted_pts, seg = _segment_inputs(idx_d.view(-1), P)
    ws = torch.empty(hip.lib().papr_segment_reduce_workspace_bytes(M) // 4, device=d)
    outs = []
    for pair in (pair_a, pair_b):
        d_pts = torch.zeros((P, 3), device=d)
        hip.check(hip.lib().papr_segment_reduce(hip.ptr(order), hip.ptr(sorted_pts), hip.ptr(seg), M, P, hip.ptr(pair), None, None, 0, 0, 0,
                                                hip.ptr(d_pts), None, None, 0, hip.ptr(ws), hip.stream_ptr()), "segment_reduce")
        outs.append(d_pts.cpu())
    ref = st["points"].grad
    scale = ref.abs().max().item()
    np.testing.assert_allclose(outs[0].numpy(), ref.numpy(), rtol=0, atol=3e-4 * scale)
    np.testing.assert_allclose(outs[0].numpy(), outs[1].numpy(), rtol=0, atol=2e-5 * scale)      # the same arithmetic in another order
    np.testing.assert_allclose(pair_a.cpu().numpy(), pair_b.cpu().numpy(), rtol=0, atol=2e-5 * pair_b.abs().max().item())


# ------------------------------------------------------------------------------------- row norm
@pytest.mark.parametrize("rows,width,ld", [(1000, 117, 120), (37, 39, 40), (513, 256, 256), (5, 600, 600)])
def test_rownorm_forward_backward(rows, width, ld):
    from papr_amd import ops
    gen = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, width, generator=gen) * 3 + 0.7
    x.requires_grad_(True)
    ones, zeros = torch.ones(width), torch.zeros(width)
    y_ref = O.custom_layernorm(x, ones, zeros, 1e-6)
    gy = torch.randn(rows, width, generator=gen)
    (y_ref * gy).sum().backward()
    xp = torch.zeros(rows, ld); xp[:, :width] = x.detach()
    xd = xp.to(dev())
    stats = ops.rownorm_(xd, width, 1e-6)
    np.testing.assert_allclose(xd.cpu()[:, :width].numpy(), y_ref.detach().numpy(), rtol=0, atol=3e-6)
    gp = torch.zeros(rows, ld); gp[:, :width] = gy
    gd = gp.to(dev())
    ops.rownorm_bwd_(gd, xd, stats, width, 1e-6)
    np.testing.assert_allclose(gd.cpu()[:, :width].numpy(), x.grad.numpy(), rtol=0, atol=3e-5 * x.grad.abs().max().item())


# ------------------------------------------------------------------------------------------- K3
def _ref_mlp(x, ws, bs, acts, skips, d_in):
    h = x
    for i, (w, b) in enumerate(zip(ws, bs)):
        if i in skips:
            h = torch.cat([h, x], -1)
        h = torch.nn.functional.linear(h, w, b)
        if acts[i] == "relu":
            h = torch.relu(h)
        elif acts[i] == "leakyrelu":
            h = torch.nn.functional.leaky_relu(h, 0.2)
    return h


@pytest.mark.parametrize("M,d_in,width,d_out,n,act,skips", [
    (1000, 117, 256, 256, 5, "relu", []),
    (777, 142, 256, 32, 8, "leakyrelu", [5]),
    (129, 39, 64, 3, 4, "relu", []),
    (4100, 142, 128, 64, 3, "relu", [1]),
    (33, 27, 256, 256, 2, "none", []),
    # 244 row slices of nine stages each in the weight-gradient kernel.  No activation: with 18 M ReLU inputs a few
    # lie within rounding of zero and the masks of two correct fp32 implementations differ (torch fp32 itself is
    # 6e-2 from torch fp64 on d_x there; scripts/probes/mlp_err.py)
    (70001, 64, 256, 160, 3, "none", []),
])
@pytest.mark.parametrize("rows", ["f32", "f16"])
def test_mlp_forward_backward_vs_torch(M, d_in, width, d_out, n, act, skips, rows, monkeypatch):
    # rows: what the fused runs keep for their weight gradients.  "f32" (PAPR_H3_ROWS=f32, mode PAPR_MLP_H3): fp32 rows, three f16 products per fp32 product
    # everywhere -- the strict bars below.  "f16" (the default since round 6, PAPR_MLP_H3_F16ROWS): forward and input gradient IDENTICAL bars, weight / bias
    # gradients from f16-rounded rows (one product): their bar is the model-level one (rms within 1.5e-4 of the tensor's maximum, conftest.grad_check)
    monkeypatch.setenv("PAPR_H3_ROWS", rows)
    wtol = 3e-5 if rows == "f32" else None
    from papr_amd import ops
    gen = torch.Generator().manual_seed(M)
    ecfg = dict(n_ff_layer=n, d_ff=width, d_ff_out=d_out, norm="none", ff_act=act, ff_last_act="none", skip_layers=skips)
    spec = ops.MlpSpec("t", d_in, ecfg)
    ws, bs = [], []
    for i in range(n):
        fi = (d_in if i == 0 else width) + (d_in if i in skips else 0)
        fo = d_out if i == n - 1 else width
        ws.append(((torch.rand(fo, fi, generator=gen) * 2 - 1) * (6.0 / (fi + fo)) ** 0.5).requires_grad_(True))
        bs.append(((torch.rand(fo, generator=gen) * 2 - 1) * 0.1).requires_grad_(True))
    x = (torch.randn(M, d_in, generator=gen)).requires_grad_(True)
    y_ref = _ref_mlp(x, ws, bs, [act] * (n - 1) + ["none"], skips, d_in)
    gy = torch.randn(M, d_out, generator=gen)
    (y_ref * gy).sum().backward()

    d = dev()
    wd = [w.detach().to(d).requires_grad_(True) for w in ws]
    bd = [b.detach().to(d).requires_grad_(True) for b in bs]
    ew, eb = ops.prepare_mlp_weights(spec, wd, bd)
    xp = torch.zeros(M, spec.ld_in); xp[:, :d_in] = x.detach()
    xd = xp.to(d)
    outs = ops.mlp_forward(spec, [w.detach() for w in ew], [b.detach() for b in eb], xd, M, keep=True)
    y = outs[-1].cpu()[:, :d_out]
    np.testing.assert_allclose(y.numpy(), y_ref.detach().numpy(), rtol=0, atol=2e-5 * max(1.0, y_ref.abs().max().item()))
    # inference mode (ping-pong buffers) gives the same numbers
    outs2 = ops.mlp_forward(spec, [w.detach() for w in ew], [b.detach() for b in eb], xd, M, keep=False)
    assert torch.equal(outs2[-1].cpu(), outs[-1].cpu())
    gp = torch.zeros(M, spec.ld_out[-1]); gp[:, :d_out] = gy
    wmax = max(width, spec.ld_in, spec.ld_out[-1])
    scratch = [torch.empty((M, wmax), device=d) for _ in range(2)]
    gp_d = gp.to(d)
    d_ws, d_bs, d_x = ops.mlp_backward(spec, [w.detach() for w in ew], [b.detach() for b in eb], xd, M, outs, gp_d, scratch, True)
    torch.cuda.synchronize()
    # chain the gradients of the effective weights back to the reference-shaped parameters
    torch.autograd.backward(ew + eb, d_ws + d_bs)
    for i in range(n):
        if wtol is None:
            for got, ref, nm in ((wd[i].grad, ws[i].grad, "dW%d" % i), (bd[i].grad, bs[i].grad, "db%d" % i)):
                # (one f16 rounding per operand element, 2^-11 relative: with a few hundred rows nothing averages -- 1e-3 rms of the tensor's maximum here; the
                #  model-level goldens hold the same gradients to 1.5e-4, tests/test_hip_model.py, unchanged)
                e = (got.cpu() - ref).abs() / (ref.abs().max().item() + 1e-12)
                assert e.pow(2).mean().sqrt().item() <= 1e-3 and e.max().item() <= 4e-3, (nm, e.pow(2).mean().sqrt().item(), e.max().item())
            continue
        sw = ws[i].grad.abs().max().item() + 1e-12
        np.testing.assert_allclose(wd[i].grad.cpu().numpy(), ws[i].grad.numpy(), rtol=0, atol=wtol * sw, err_msg="dW%d" % i)
        sb = bs[i].grad.abs().max().item() + 1e-12
        np.testing.assert_allclose(bd[i].grad.cpu().numpy(), bs[i].grad.numpy(), rtol=0, atol=wtol * sb, err_msg="db%d" % i)
    sx = x.grad.abs().max().item()
    np.testing.assert_allclose(d_x.cpu()[:, :d_in].numpy(), x.grad.numpy(), rtol=0, atol=3e-5 * sx)
    # without the state of the forward call (row maxima, sign words) the backward pass takes its other route -- fp32
    # activation rows as masks, fp32 MFMA weight-gradients -- and must land on the same gradients.  (fp32 rows only: with f16 rows the inner rows
    # of a fused run ARE part of that state -- include/papr_hip.h, PAPR_MLP_H3_F16ROWS)
    if rows == "f16":
        return
    d_ws2, d_bs2, d_x2 = ops.mlp_backward(spec, [w.detach() for w in ew], [b.detach() for b in eb], xd, M, list(outs), gp.to(d), scratch, True)
    torch.cuda.synchronize()
    for i in range(n):
        sw = d_ws[i].abs().max().item() + 1e-12
        np.testing.assert_allclose(d_ws2[i].cpu().numpy(), d_ws[i].cpu().numpy(), rtol=0, atol=3e-5 * sw, err_msg="dW%d (no saved state)" % i)
    np.testing.assert_allclose(d_x2.cpu().numpy(), d_x.cpu().numpy(), rtol=0, atol=3e-5 * sx)


# (4000 x 20 rows: workgroups carry more than one pair of tiles, so the staging slots with the early request of the next rows run too)
@pytest.mark.parametrize("R,k,d_in,width,d_out,n", [(1234, 20, 117, 256, 256, 5), (61, 7, 39, 256, 64, 3), (300, 20, 27, 64, 32, 2), (5, 63, 117, 256, 256, 5),
                                                    (4000, 20, 117, 256, 256, 5)])
def test_row_dots_from_the_last_row_phase(R, k, d_in, width, d_out, n):
    """papr_row_norm.dots: the score dot products leave the fused key run instead of the key embedding (inference), and equal what
    the standardised rows give when they are written and multiplied afterwards (papr_row_dots, torch).  The (300, .., 64, 32) chain is
    too narrow for a fused run: there the library's own fallback (rows, then papr_row_dots) answers."""
    from papr_amd import hip, ops
    import ctypes as C
    gen = torch.Generator().manual_seed(R)
    M = R * k
    ecfg = dict(n_ff_layer=n, d_ff=width, d_ff_out=d_out, norm="none", ff_act="relu", ff_last_act="none", skip_layers=[])
    spec = ops.MlpSpec("t", d_in, ecfg)
    d = dev()
    ws = [((torch.rand(d_out if i == n - 1 else width, d_in if i == 0 else width, generator=gen) * 2 - 1) * 0.15).to(d) for i in range(n)]
    bs = [((torch.rand(d_out if i == n - 1 else width, generator=gen) * 2 - 1) * 0.1).to(d) for i in range(n)]
    ew, eb = ops.prepare_mlp_weights(spec, ws, bs)
    xp = torch.zeros(M, spec.ld_in); xp[:, :d_in] = torch.randn(M, d_in, generator=gen)
    g = torch.zeros(R, d_out + 4); g[:, :d_out] = torch.randn(R, d_out, generator=gen)
    gd = g.to(d)
    eps = 1e-6
    plain = ops.mlp_forward(spec, ew, eb, xp.to(d), M, keep=False, out_norm=(d_out, eps))
    rows = plain[-1].clone()
    want = (rows[:, :d_out].double().view(R, k, d_out) * gd[:, None, :d_out].double()).sum(-1).view(-1)
    tol = 2e-6 * float((rows[:, :d_out].double().norm(dim=1).max() * gd.double().norm(dim=1).max()))
    own = torch.empty(M, device=d)
    hip.check(hip.lib().papr_row_dots(hip.ptr(rows), M, d_out, rows.stride(0), hip.ptr(gd), gd.stride(0), k, hip.ptr(own), hip.stream_ptr()), "papr_row_dots")
    np.testing.assert_allclose(own.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=tol)
    inf = ops.mlp_forward(spec, ew, eb, xp.to(d), M, keep=False, out_norm=(d_out, eps), dot_rows=gd, rows_per_dot=k)
    np.testing.assert_allclose(inf.dots.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=tol)
    assert torch.equal(inf.norm_stats, plain.norm_stats)
    # training form: rows AND dots; the rows are the ones a call without dots writes, the dots the inference call's
    kept = ops.mlp_forward(spec, ew, eb, xp.to(d), M, keep=True, out_norm=(d_out, eps), dot_rows=gd, rows_per_dot=k)
    assert torch.equal(kept[-1], rows) and torch.equal(kept.dots, inf.dots)
    # every row stands alone: a chunk of the rows gives the same bits
    M2 = (R // 2) * k
    if M2:
        part = ops.mlp_forward(spec, ew, eb, xp[:M2].to(d), M2, keep=False, out_norm=(d_out, eps), dot_rows=gd, rows_per_dot=k)
        assert torch.equal(part.dots, inf.dots[:M2])
    # raw rows (papr_row_norm.raw_mean, ABI 25; fused runs only): the last output un-standardised + the rows' means; standardising them with the
    # two instructions the run would have used gives the bits of the standardised rows; statistics and dots unchanged
    fused_run = n >= 2 and all(L["n_out"] % 32 == 0 for L in spec.layers)
    if fused_run:
        raw = ops.mlp_forward(spec, ew, eb, xp.to(d), M, keep=True, out_norm=(d_out, eps), dot_rows=gd, rows_per_dot=k, raw_rows=True)
        assert raw.norm_mean is not None and torch.equal(raw.norm_stats, kept.norm_stats) and torch.equal(raw.dots, kept.dots)
        assert torch.equal((raw[-1][:, :d_out] - raw.norm_mean[:, None]) * raw.norm_stats[:, :1], rows[:, :d_out])
        nonorm = ops.mlp_forward(spec, ew, eb, xp.to(d), M, keep=False)
        assert torch.equal(raw[-1][:, :d_out], nonorm[-1][:, :d_out])
        np.testing.assert_allclose(raw.norm_mean.cpu().numpy(), nonorm[-1][:, :d_out].double().mean(1).cpu().numpy(), rtol=0, atol=1e-5)
    else:
        with pytest.raises(RuntimeError):           # no silent fallback: asked of a chain that is no fused run, the call fails
            ops.mlp_forward(spec, ew, eb, xp.to(d), M, keep=True, out_norm=(d_out, eps), dot_rows=gd, rows_per_dot=k, raw_rows=True)


@pytest.mark.parametrize("M,d_in,width,d_out,n", [(24680, 117, 256, 256, 5), (1220, 39, 256, 64, 3), (6000, 27, 64, 32, 2), (45, 117, 256, 256, 5)])
def test_in_norm_with_given_statistics(M, d_in, width, d_out, n):
    """papr_row_norm.given_mean (ABI 24): the LayerNorm core in front of an MLP with statistics the CALLER hands over -- mean, std, 1 / (std + eps)
    of every input row, here taken by torch in float64 -- against the same call letting the library take them, and against torch: the standardised
    rows it leaves in x (training form) and the MLP's output.  The narrow chain runs without a fused run (the library applies the statistics in a
    pass of its own)."""
    from papr_amd import ops
    gen = torch.Generator().manual_seed(M)
    ecfg = dict(n_ff_layer=n, d_ff=width, d_ff_out=d_out, norm="none", ff_act="relu", ff_last_act="none", skip_layers=[])
    spec = ops.MlpSpec("t", d_in, ecfg)
    d = dev()
    ws = [((torch.rand(d_out if i == n - 1 else width, d_in if i == 0 else width, generator=gen) * 2 - 1) * 0.15).to(d) for i in range(n)]
    bs = [((torch.rand(d_out if i == n - 1 else width, generator=gen) * 2 - 1) * 0.1).to(d) for i in range(n)]
    ew, eb = ops.prepare_mlp_weights(spec, ws, bs)
    xp = torch.zeros(M, spec.ld_in)
    xp[:, :d_in] = torch.randn(M, d_in, generator=gen) * 2.0 + 0.3
    eps = 1e-6
    x64 = xp[:, :d_in].double()
    mean, std = x64.mean(1), x64.std(1)
    stats = torch.stack([1.0 / (std + eps), std], 1).float().to(d)
    x_own, x_given = xp.to(d), xp.to(d)
    own = ops.mlp_forward(spec, ew, eb, x_own, M, keep=True, in_norm=(d_in, eps))
    given = ops.mlp_forward(spec, ew, eb, x_given, M, keep=True, in_norm=(d_in, eps, stats, mean.float().to(d)))
    want_x = ((x64 - mean[:, None]) / (std[:, None] + eps)).float()
    np.testing.assert_allclose(x_given.cpu()[:, :d_in].numpy(), want_x.numpy(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(x_own.cpu()[:, :d_in].numpy(), want_x.numpy(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(own.in_stats.cpu().numpy(), stats.cpu().numpy(), rtol=3e-6, atol=0)
    scale = float(given[-1].abs().max())
    np.testing.assert_allclose(given[-1].cpu().numpy(), own[-1].cpu().numpy(), rtol=0, atol=2e-5 * scale)
    assert given.in_stats is stats or torch.equal(given.in_stats, stats)


# ------------------------------------------------------------------------------------------- K4
@pytest.mark.parametrize("R,k,d_model,Cc,act,normalize", [(300, 20, 256, 32, "relu", True), (65, 12, 64, 3, "relu", True),
                                                          (17, 1, 256, 32, "none", False), (40, 63, 128, 32, "leakyrelu", True)])
def test_attention_tail_forward_backward(R, k, d_model, Cc, act, normalize):
    from papr_amd import hip
    gen = torch.Generator().manual_seed(R * k)
    P = 500
    kp = (torch.randn(R, k, d_model, generator=gen) * 0.7).requires_grad_(True)
    qp = (torch.randn(R, 1, d_model, generator=gen) * 0.7).requires_grad_(True)
    ldv = (Cc + 31) // 32 * 32
    v = torch.randn(R, k, Cc, generator=gen).requires_grad_(True)
    influ = (torch.rand(P, 1, generator=gen) * 1.2 - 0.2).requires_grad_(True)
    idx = torch.randint(0, P, (R, k), generator=gen)
    bkg = 3.0
    sb = torch.randn(R, generator=gen).requires_grad_(True)          # per-ray additive term of the dot products
    scale_dim = 2 * d_model if R == 65 else d_model                  # the divisor need not be the dot width
    sc = (torch.matmul(qp, kp.transpose(-2, -1)).squeeze(1) + sb[:, None]) / scale_dim ** 0.5
    sc = O._act(sc, act)
    fused_ref, attn_ref = O.attention_tail(sc, influ[idx].squeeze(-1), v, bkg, normalize)
    gf = torch.randn(R, Cc, generator=gen)
    ga = torch.randn(R, k + 1, generator=gen)
    ((fused_ref * gf).sum() + (attn_ref * ga).sum()).backward()

    d = dev()
    td = hip.TailDesc()
    td.k, td.d_model, td.C, td.ld_kp, td.ld_qp, td.ld_v = k, d_model, Cc, d_model, d_model, ldv
    td.score_act, td.normalize, td.bkg_score, td.scale_dim = hip.ACT[act], int(normalize), bkg, scale_dim
    sbd = sb.detach().to(d)
    kpd = kp.detach().reshape(R * k, d_model).to(d)
    qpd = qp.detach().reshape(R, d_model).to(d)
    vp = torch.zeros(R * k, ldv); vp[:, :Cc] = v.detach().reshape(R * k, Cc)
    vd = vp.to(d)
    infd, idxd = influ.detach().to(d), idx.int().to(d)
    scores = torch.empty((R, k), device=d); attn = torch.empty((R, k + 1), device=d); fused = torch.empty((R, Cc), device=d)
    hip.check(hip.lib().papr_attn_tail_fwd(C.byref(td), hip.ptr(kpd), hip.ptr(qpd), hip.ptr(sbd), hip.ptr(vd), hip.ptr(infd), hip.ptr(idxd), R,
                                           hip.ptr(scores), hip.ptr(attn), hip.ptr(fused), hip.stream_ptr()), "tail_fwd")
    np.testing.assert_allclose(scores.cpu().numpy(), sc.detach().numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(attn.cpu().numpy(), attn_ref.detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(fused.cpu().numpy(), fused_ref.detach().numpy(), rtol=0, atol=1e-5)
    d_kp = torch.empty_like(kpd); d_qp = torch.empty_like(qpd); d_v = torch.empty_like(vd)
    d_inf = torch.zeros((P, 1), device=d)
    d_sb = torch.empty((R,), device=d)
    gf_d, ga_d = gf.to(d), ga.to(d)
    hip.check(hip.lib().papr_attn_tail_bwd(C.byref(td), hip.ptr(kpd), hip.ptr(qpd), hip.ptr(vd), hip.ptr(infd), hip.ptr(idxd), R,
                                           hip.ptr(scores), hip.ptr(attn), hip.ptr(gf_d), hip.ptr(ga_d), hip.ptr(d_kp),
                                           hip.ptr(d_qp), hip.ptr(d_v), hip.ptr(d_inf), hip.ptr(d_sb), None, None, None, None, None, None, hip.stream_ptr()), "tail_bwd")
    torch.cuda.synchronize()
    if d_model == 256 and Cc == 32:
        # the same gradient rows in the fused runs' input format (papr_f16_rows; both forms): bit for bit what the runs' staging makes of the fp32 rows
        import f16_rows
        from papr_amd import ops
        for split in (False, True):
            k16, v16 = ops.F16Rows(d, R * k, 256, split), ops.F16Rows(d, R * k, 32, split)
            d_qp2 = torch.empty_like(d_qp); d_inf2 = torch.zeros((P, 1), device=d); d_sb2 = torch.empty((R,), device=d)
            hip.check(hip.lib().papr_attn_tail_bwd(C.byref(td), hip.ptr(kpd), hip.ptr(qpd), hip.ptr(vd), hip.ptr(infd), hip.ptr(idxd), R,
                                                   hip.ptr(scores), hip.ptr(attn), hip.ptr(gf_d), hip.ptr(ga_d), None,
                                                   hip.ptr(d_qp2), None, hip.ptr(d_inf2), hip.ptr(d_sb2), None, None, None, None, k16.ref(), v16.ref(), hip.stream_ptr()), "tail_bwd")
            torch.cuda.synchronize()
            assert torch.equal(d_qp2, d_qp) and torch.equal(d_sb2, d_sb)
            for rows, got in ((d_kp, k16), (d_v[:, :32], v16)):
                hi, lo, inv, scale, mx = f16_rows.expected(rows, split)
                assert torch.equal(got.tables[2], mx) and torch.equal(got.tables[1], scale) and torch.equal(got.tables[0], inv)
                assert torch.equal(got.hi, hi) and (not split or torch.equal(got.lo, lo))
                assert (got.float() - rows).abs().max() <= (2e-7 if split else 1e-3) * rows.abs().max()

    tol = lambda ref: 3e-5 * ref.abs().max().item() + 1e-9
    np.testing.assert_allclose(d_kp.cpu().numpy(), kp.grad.reshape(R * k, -1).numpy(), rtol=0, atol=tol(kp.grad))
    np.testing.assert_allclose(d_qp.cpu().numpy(), qp.grad.reshape(R, -1).numpy(), rtol=0, atol=tol(qp.grad))
    np.testing.assert_allclose(d_v.cpu()[:, :Cc].numpy(), v.grad.reshape(R * k, -1).numpy(), rtol=0, atol=tol(v.grad))
    assert torch.all(d_v.cpu()[:, Cc:] == 0)
    np.testing.assert_allclose(d_inf.cpu().numpy(), influ.grad.numpy(), rtol=0, atol=tol(influ.grad))
    np.testing.assert_allclose(d_sb.cpu().numpy(), sb.grad.numpy(), rtol=0, atol=tol(sb.grad))


@pytest.mark.parametrize("R,k,d_model,act", [(130, 20, 256, "relu"), (33, 7, 64, "leakyrelu"), (20, 3, 128, "none")])
def test_attention_tail_backward_through_standardised_keys(R, k, d_model, act):
    """kp_norm_stats: the key rows are LayerNorm-core outputs (models/attn.py:39-42) and d_kp comes back as the gradient
    w.r.t. the rows in front of the standardisation -- against autograd through the oracle's row norm."""
    from papr_amd import hip
    gen = torch.Generator().manual_seed(R + k)
    P, Cc, ldv, eps, bkg = 200, 32, 32, 1e-6, 5.0
    x = (torch.randn(R, k, d_model, generator=gen) * 1.3 + 0.2).requires_grad_(True)
    mean = x.mean(-1, keepdim=True)
    sigma = x.std(-1, keepdim=True)                                  # unbiased, eps on the std (attn.py:41-42)
    kp = (x - mean) / (sigma + eps)
    qp = (torch.randn(R, 1, d_model, generator=gen) * 0.2).requires_grad_(True)
    v = torch.randn(R, k, Cc, generator=gen)
    influ = torch.rand(P, 1, generator=gen) * 1.2 - 0.2
    idx = torch.randint(0, P, (R, k), generator=gen)
    sb = torch.randn(R, generator=gen) * 0.5
    sc = O._act((torch.matmul(qp, kp.transpose(-2, -1)).squeeze(1) + sb[:, None]) / d_model ** 0.5, act)
    fused_ref, attn_ref = O.attention_tail(sc, influ[idx].squeeze(-1), v, bkg, True)
    gf = torch.randn(R, Cc, generator=gen)
    ga = torch.randn(R, k + 1, generator=gen)
    ((fused_ref * gf).sum() + (attn_ref * ga).sum()).backward()

    d = dev()
    td = hip.TailDesc()
    td.k, td.d_model, td.C, td.ld_kp, td.ld_qp, td.ld_v = k, d_model, Cc, d_model, d_model, ldv
    td.score_act, td.normalize, td.bkg_score, td.scale_dim = hip.ACT[act], 1, bkg, d_model
    kpd = kp.detach().reshape(R * k, d_model).contiguous().to(d)
    stats = torch.stack([1.0 / (sigma.detach().reshape(-1) + eps), sigma.detach().reshape(-1)], 1).contiguous().to(d)
    qpd, vd, sbd = qp.detach().reshape(R, d_model).to(d), v.reshape(R * k, Cc).contiguous().to(d), sb.to(d)
    infd, idxd = influ.to(d), idx.int().to(d)
    scores = torch.empty((R, k), device=d); attn = torch.empty((R, k + 1), device=d); fused = torch.empty((R, Cc), device=d)
    hip.check(hip.lib().papr_attn_tail_fwd(C.byref(td), hip.ptr(kpd), hip.ptr(qpd), hip.ptr(sbd), hip.ptr(vd), hip.ptr(infd), hip.ptr(idxd), R,
                                           hip.ptr(scores), hip.ptr(attn), hip.ptr(fused), hip.stream_ptr()), "tail_fwd")
    d_x = torch.empty_like(kpd); d_qp = torch.empty_like(qpd); d_v = torch.empty_like(vd)
    d_inf = torch.zeros((P, 1), device=d)
    gf_d, ga_d = gf.to(d), ga.to(d)                                  # (kept alive: a temporary's block would be handed to the next one)
    hip.check(hip.lib().papr_attn_tail_bwd(C.byref(td), hip.ptr(kpd), hip.ptr(qpd), hip.ptr(vd), hip.ptr(infd), hip.ptr(idxd), R,
                                           hip.ptr(scores), hip.ptr(attn), hip.ptr(gf_d), hip.ptr(ga_d), hip.ptr(d_x),
                                           hip.ptr(d_qp), hip.ptr(d_v), hip.ptr(d_inf), None, None, hip.ptr(stats), hip.ptr(sbd), None,
                                           None, None, hip.stream_ptr()), "tail_bwd")
    torch.cuda.synchronize()
    tol = lambda ref: 1e-4 * ref.abs().max().item() + 1e-9
    np.testing.assert_allclose(d_x.cpu().numpy(), x.grad.reshape(R * k, -1).numpy(), rtol=0, atol=tol(x.grad))
    np.testing.assert_allclose(d_qp.cpu().numpy(), qp.grad.reshape(R, -1).numpy(), rtol=0, atol=tol(qp.grad))
    # kp_mean (ABI 25): the key rows RAW, standardised as they are read -- the same bits as from the standardised rows.  Raw rows whose
    # (x - mean) * rinv reproduces kpd exactly: x = kpd / rinv + mean need not round-trip, so build them the other way round
    raw = (torch.randn(R * k, d_model, generator=torch.Generator().manual_seed(R)) * 2.0 + 0.3).to(d)
    mean = raw.mean(1)
    st2 = torch.stack([1.0 / (raw.std(1) + 1e-6), raw.std(1)], 1).contiguous()
    std_rows = ((raw - mean[:, None]) * st2[:, :1]).contiguous()
    outs = []
    for rows, mptr in ((std_rows, None), (raw, hip.ptr(mean))):
        hip.check(hip.lib().papr_attn_tail_fwd(C.byref(td), hip.ptr(std_rows), hip.ptr(qpd), hip.ptr(sbd), hip.ptr(vd), hip.ptr(infd), hip.ptr(idxd), R,
                                               hip.ptr(scores), hip.ptr(attn), hip.ptr(fused), hip.stream_ptr()), "tail_fwd")
        a, b, c_ = torch.empty_like(kpd), torch.empty_like(qpd), torch.empty_like(vd)
        hip.check(hip.lib().papr_attn_tail_bwd(C.byref(td), hip.ptr(rows), hip.ptr(qpd), hip.ptr(vd), hip.ptr(infd), hip.ptr(idxd), R,
                                               hip.ptr(scores), hip.ptr(attn), hip.ptr(gf_d), hip.ptr(ga_d), hip.ptr(a),
                                               hip.ptr(b), hip.ptr(c_), hip.ptr(d_inf), None, None, hip.ptr(st2), hip.ptr(sbd), mptr,
                                               None, None, hip.stream_ptr()), "tail_bwd")
        outs.append((a.clone(), b.clone(), c_.clone()))
    torch.cuda.synchronize()
    for u, w_ in zip(outs[0], outs[1]):
        assert torch.equal(u, w_), "raw key rows standardised on the fly must give the bits of standardised rows"


@pytest.mark.parametrize("n_out,n_in,ld_eff", [(256, 117, 120), (256, 256, 256), (32, 39, 40), (5, 3, 8)])
def test_layernorm_affine_fold(n_out, n_in, ld_eff):
    """papr_ln_fold_fwd / bwd against the torch expressions they replace: W (a_2 xh + b_2) + c = (W a_2) xh + (W b_2 + c)."""
    from papr_amd import ops
    gen = torch.Generator().manual_seed(n_out + n_in)
    w = torch.randn(n_out, n_in, generator=gen).requires_grad_(True)
    c = torch.randn(n_out, generator=gen).requires_grad_(True)
    a2 = (1 + 0.3 * torch.randn(n_in, generator=gen)).requires_grad_(True)
    b2 = (0.3 * torch.randn(n_in, generator=gen)).requires_grad_(True)
    gw, gb = torch.randn(n_out, ld_eff, generator=gen), torch.randn(n_out, generator=gen)
    eff_w = torch.nn.functional.pad(w * a2, (0, ld_eff - n_in))
    eff_b = c + (w * b2).sum(1)
    ((eff_w * gw).sum() + (eff_b * gb).sum()).backward()
    d = dev()
    leaves = [t.detach().to(d).requires_grad_(True) for t in (w, c, a2, b2)]
    ew, eb = ops._LnFoldFn.apply(*leaves, ld_eff)
    ((ew * gw.to(d)).sum() + (eb * gb.to(d)).sum()).backward()
    np.testing.assert_allclose(ew.detach().cpu().numpy(), eff_w.detach().numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(eb.detach().cpu().numpy(), eff_b.detach().numpy(), rtol=0, atol=2e-6 * (1 + eff_b.abs().max().item()))
    for got, ref in zip(leaves, (w, c, a2, b2)):
        np.testing.assert_allclose(got.grad.cpu().numpy(), ref.grad.numpy(), rtol=0, atol=3e-6 * (1 + ref.grad.abs().max().item()))


@pytest.mark.parametrize("B,H,W,c_in,c_out,relu", [(1, 20, 24, 32, 128, True), (2, 9, 7, 64, 96, False), (1, 40, 40, 256, 512, True),
                                                     (1, 13, 130, 128, 128, True)])
def test_conv3x3_forward_and_gradients_match_torch(B, H, W, c_in, c_out, relu):
    """papr_conv3x3_fwd (forward and, through the flipped weight, data-gradient) against torch.nn.functional.conv2d in fp32
    on the CPU -- the 3x3 layers of the reference's SmallUNet (models/unet.py:16-33)."""
    from papr_amd import ops
    gen = torch.Generator().manual_seed(B * H + c_out)
    x = torch.randn(B, c_in, H, W, generator=gen).requires_grad_(True)
    w = (torch.randn(c_out, c_in, 3, 3, generator=gen) * (2.0 / (9 * c_in)) ** 0.5).requires_grad_(True)
    b = (torch.randn(c_out, generator=gen) * 0.1).requires_grad_(True)
    y = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    y = torch.relu(y) if relu else y
    gy = torch.randn(B, c_out, H, W, generator=gen)
    (y * gy.double()).sum().backward()
    d = dev()
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(d).requires_grad_(True)
    wd, bd = w.detach().to(d).requires_grad_(True), b.detach().to(d).requires_grad_(True)
    yd = ops._Conv3x3Fn.apply(xd, wd, bd, relu)
    (yd * gy.permute(0, 2, 3, 1).contiguous().to(d)).sum().backward()
    tol = lambda ref: 2e-6 * ref.abs().max().item()
    np.testing.assert_allclose(yd.detach().cpu().permute(0, 3, 1, 2).numpy(), y.detach().float().numpy(), rtol=0, atol=tol(y))
    np.testing.assert_allclose(xd.grad.cpu().permute(0, 3, 1, 2).numpy(), x.grad.numpy(), rtol=0, atol=3e-6 * x.grad.abs().max().item())
    np.testing.assert_allclose(wd.grad.cpu().numpy(), w.grad.numpy(), rtol=0, atol=2e-5 * w.grad.abs().max().item())
    np.testing.assert_allclose(bd.grad.cpu().numpy(), b.grad.numpy(), rtol=0, atol=2e-5 * b.grad.abs().max().item())


def test_conv3x3_gradient_without_an_own_form_raises_by_name():
    """No silent aten / MIOpen fallback in the render head (round 6): the forward kernel takes any output width that is a multiple of 4, the
    data-gradient kernel needs a multiple of 32 -- a gradient it has no form for raises by name instead of going to aten.convolution_backward
    (which stays reachable only behind PAPR_DEBUG_TORCH_HEAD=wgrad, papr_amd/debug.py)."""
    from papr_amd import ops
    d = dev()
    x = torch.randn(1, 6, 5, 64, device=d, requires_grad=True)
    w = torch.randn(36, 64, 3, 3, device=d, requires_grad=True)
    b = torch.zeros(36, device=d, requires_grad=True)
    y = ops._Conv3x3Fn.apply(x, w, b, False)
    assert y.shape == (1, 6, 5, 36)
    with pytest.raises(NotImplementedError, match="data gradient of Conv2d\\(64, 36, 3x3\\)"):
        y.sum().backward()


def test_small_unet_layer_outside_the_own_kernels_raises_on_the_device():
    """... and a layer of a head whose shape the own kernels do not cover raises when it meets a device tensor (rounds 3-5: nn.Conv2d / MaxPool2d /
    ConvTranspose2d took it without a word)."""
    from papr_amd.unet import ConvStage, DownStage, Head, UpStage
    d = dev()
    with pytest.raises(NotImplementedError, match="Conv2d\\(24, 32, 3x3\\)"):
        ConvStage(24, 32).to(d)(torch.randn(1, 24, 8, 8, device=d))
    with pytest.raises(NotImplementedError, match="MaxPool2d"):
        DownStage(6, 32).to(d)(torch.randn(1, 6, 8, 8, device=d))
    with pytest.raises(NotImplementedError, match="ConvTranspose2d\\(96, 48, 2x2\\)"):
        UpStage(96, 32).to(d)(torch.randn(1, 96, 4, 4, device=d), torch.randn(1, 48, 8, 8, device=d))
    with pytest.raises(NotImplementedError, match="Conv2d\\(128, 7, 1x1\\)"):
        Head(128, 7).to(d)(torch.randn(1, 128, 8, 8, device=d))
    with pytest.raises(NotImplementedError, match="float16"):
        ConvStage(32, 32).to(d).half()(torch.randn(1, 32, 8, 8, device=d).half())


@pytest.mark.parametrize("B,H,W,C", [(1, 40, 40, 128), (2, 7, 9, 4), (1, 20, 20, 256), (3, 2, 5, 36)])
def test_maxpool2_matches_torch_exactly(B, H, W, C):
    """papr_maxpool2_fwd / _bwd against torch.nn.functional.max_pool2d on the CPU, bit for bit, on a map with many ties (a
    ReLU output, like the maps the reference pools: models/unet.py:36-49) and with odd sizes (last row / column dropped)."""
    from papr_amd import ops
    gen = torch.Generator().manual_seed(B * H * W + C)
    x = torch.relu(torch.randn(B, C, H, W, generator=gen)).requires_grad_(True)
    y = torch.nn.functional.max_pool2d(x, 2)
    gy = torch.randn(y.shape, generator=gen)
    (y * gy).sum().backward()
    d = dev()
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(d).requires_grad_(True)
    yd = ops._MaxPool2Fn.apply(xd)
    (yd * gy.permute(0, 2, 3, 1).contiguous().to(d)).sum().backward()
    assert torch.equal(yd.detach().cpu().permute(0, 3, 1, 2), y.detach())
    assert torch.equal(xd.grad.cpu().permute(0, 3, 1, 2), x.grad)
    with torch.no_grad():
        assert torch.equal(ops._MaxPool2Fn.apply(xd.detach()), yd.detach())


@pytest.mark.parametrize("B,H,W,c_in,c_out", [(1, 10, 10, 512, 256), (1, 20, 20, 256, 128), (2, 5, 7, 64, 64), (1, 33, 50, 128, 64)])
def test_upconv2x2_forward_and_gradients_match_torch(B, H, W, c_in, c_out):
    """papr_upconv2x2_fwd / _dgrad / _wgrad against torch.nn.functional.conv_transpose2d(stride 2) in float64 on the CPU -- the
    upsampling layers of the reference's SmallUNet (models/unet.py:62); same bars as the 3x3 layers."""
    from papr_amd import ops
    gen = torch.Generator().manual_seed(B * H + c_out)
    x = torch.randn(B, c_in, H, W, generator=gen).requires_grad_(True)
    w = (torch.randn(c_in, c_out, 2, 2, generator=gen) * (1.0 / c_in) ** 0.5).requires_grad_(True)
    b = (torch.randn(c_out, generator=gen) * 0.1).requires_grad_(True)
    y = torch.nn.functional.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)
    gy = torch.randn(B, c_out, 2 * H, 2 * W, generator=gen) * 1e-3
    (y * gy.double()).sum().backward()
    d = dev()
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(d).requires_grad_(True)
    wd = w.detach().to(d).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.detach().to(d).requires_grad_(True)
    yd = ops._UpConv2x2Fn.apply(xd, wd, bd)
    (yd * gy.permute(0, 2, 3, 1).contiguous().to(d)).sum().backward()
    np.testing.assert_allclose(yd.detach().cpu().permute(0, 3, 1, 2).numpy(), y.detach().float().numpy(), rtol=0, atol=2e-6 * y.abs().max().item())
    np.testing.assert_allclose(xd.grad.cpu().permute(0, 3, 1, 2).numpy(), x.grad.numpy(), rtol=0, atol=3e-6 * x.grad.abs().max().item())
    np.testing.assert_allclose(wd.grad.cpu().numpy(), w.grad.numpy(), rtol=0, atol=2e-5 * w.grad.abs().max().item())
    np.testing.assert_allclose(bd.grad.cpu().numpy(), b.grad.numpy(), rtol=0, atol=2e-5 * b.grad.abs().max().item())
    # a weight that is not channels-last (a freshly constructed module's) gives the same numbers
    w2 = w.detach().to(d).requires_grad_(True)
    assert torch.equal(ops._UpConv2x2Fn.apply(xd.detach(), w2, bd.detach()), yd.detach())


@pytest.mark.parametrize("B,H,W,c_in,c_out", [(1, 40, 40, 128, 3), (2, 9, 11, 32, 4), (1, 3, 5, 256, 1)])
def test_conv1x1_forward_and_gradients_match_torch(B, H, W, c_in, c_out):
    """papr_conv1x1_fwd / _bwd against torch.nn.functional.conv2d(kernel 1) in float64 on the CPU -- the output layer of the
    reference's SmallUNet (models/unet.py:86-93)."""
    from papr_amd import ops
    gen = torch.Generator().manual_seed(B * H + c_in)
    x = torch.randn(B, c_in, H, W, generator=gen).requires_grad_(True)
    w = (torch.randn(c_out, c_in, 1, 1, generator=gen) * (1.0 / c_in) ** 0.5).requires_grad_(True)
    b = (torch.randn(c_out, generator=gen) * 0.1).requires_grad_(True)
    y = torch.nn.functional.conv2d(x.double(), w.double(), b.double())
    gy = torch.randn(B, c_out, H, W, generator=gen)
    (y * gy.double()).sum().backward()
    d = dev()
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(d).requires_grad_(True)
    wd, bd = w.detach().to(d).requires_grad_(True), b.detach().to(d).requires_grad_(True)
    yd = ops._Conv1x1Fn.apply(xd, wd, bd)
    (yd * gy.permute(0, 2, 3, 1).contiguous().to(d)).sum().backward()
    np.testing.assert_allclose(yd.detach().cpu().permute(0, 3, 1, 2).numpy(), y.detach().float().numpy(), rtol=0, atol=1e-6 * y.abs().max().item())
    np.testing.assert_allclose(xd.grad.cpu().permute(0, 3, 1, 2).numpy(), x.grad.numpy(), rtol=0, atol=1e-6 * x.grad.abs().max().item())
    np.testing.assert_allclose(wd.grad.cpu().numpy(), w.grad.numpy(), rtol=0, atol=3e-6 * w.grad.abs().max().item())
    np.testing.assert_allclose(bd.grad.cpu().numpy(), b.grad.numpy(), rtol=0, atol=3e-6 * b.grad.abs().max().item())


@pytest.mark.parametrize("hw,batch,whole", [((40, 40), 1, True), ((24, 36), 1, True), ((40, 40), 1, False), ((24, 36), 1, False), ((26, 37), 1, True), ((32, 20), 2, True),
                                            ((160, 160), 1, True), ((160, 160), 1, False)])
@pytest.mark.parametrize("use_amp", [False, True])
def test_small_unet_on_own_kernels_matches_torch_module(hw, batch, whole, use_amp, monkeypatch):
    """The whole SmallUNet (reference models/unet.py:182-258) on this library's kernels against the same module run by torch
    in float64 on the CPU: output and every parameter / input gradient; and no layer of the device run may go through
    aten / MIOpen convolution or pooling (profiler check of the launched kernels).  use_amp (what every shipped scene file sets;
    the reference autocasts the module to fp16 there, models/unet.py:212): the same kernels, the same bars -- the head is not handed
    to torch autocast + MIOpen (one f16 product per fp32 product on the whole-network path: see the bars below)."""
    import copy
    import papr_amd.unet as unet_mod
    from papr_amd.unet import SmallUNet
    # whole: the network as one library call each way (papr_small_unet_fwd / _bwd: H, W multiples of 4); otherwise layer by layer (one autograd
    # function per layer; what (26, 37) takes whatever the switch says)
    monkeypatch.setattr(unet_mod, "_WHOLE_NET", whole)
    torch.manual_seed(3)
    net = SmallUNet(32, 3, use_amp=use_amp)
    x = torch.randn(batch, 32, *hw)
    gy = torch.randn(batch, 3, *hw) * 1e-2
    ref = copy.deepcopy(net).double()
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    (yr * gy.double()).sum().backward()
    d = dev()
    net_d = net.to(d)
    xd = x.to(d).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        yd = net_d(xd)
        (yd * gy.to(d)).sum().backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    foreign = [n for n in names if any(t in n.lower() for t in ("igemm", "miopen", "max_pool", "naive_conv", "cijk_"))]
    assert not foreign, foreign
    if whole and hw[0] % 4 == 0 and hw[1] % 4 == 0:      # one call each way: no concatenation copy, no ReLU-mask launch, one weight-split launch
        assert any("unet_prep" in n for n in names) and not any("conv_prep" in n or "CatArray" in n or "threshold" in n for n in names), names
    # use_amp on the whole-network path: ONE f16 product per fp32 product (operands rounded to 11 bits, fp32 accumulation: the arithmetic of the
    # reference's fp16 autocast, which also rounds the maps) -- bars 100 x the fp32-parity ones, the measured errors sit at ~1/3 of them
    one = use_amp and whole and hw[0] % 4 == 0 and hw[1] % 4 == 0
    amp = 100.0 if one else 1.0
    np.testing.assert_allclose(yd.detach().cpu().numpy(), yr.detach().float().numpy(), rtol=0, atol=amp * 1e-5 * yr.abs().max().item())
    gx, gx_ref = xd.grad.cpu().numpy(), xr.grad.float().numpy()
    if one:
        # gradients: a ReLU pre-activation within the f16 operand noise of zero (~1e-3 of the units) falls the other way and moves its whole
        # contribution (the reference's own autocast run does the same: G17 holds the build to ITS distance from fp32); rms bars, the maximum loose
        err = np.abs(gx - gx_ref)
        top = xr.grad.abs().max().item()
        worst = max((np.sqrt(((pd.grad.cpu().numpy() - pr.grad.float().numpy()) ** 2).mean()) / pr.grad.abs().max().item(), name)
                    for (name, pd), pr in zip(net_d.named_parameters(), ref.parameters()))
        print("one-product U-Net: out err %.2e of max, d_x err max %.2e rms %.2e of max, worst parameter-gradient rms %.2e of its max (%s)"
              % (np.abs(yd.detach().cpu().numpy() - yr.detach().float().numpy()).max() / yr.abs().max().item(), err.max() / top, np.sqrt((err ** 2).mean()) / top, worst[0], worst[1]))
        assert err.max() <= 0.25 * top and np.sqrt((err ** 2).mean()) <= 1e-2 * top
        assert worst[0] <= 3e-2, worst          # (measured 0.8 - 1.2e-2: bias gradients, plain sums of gradient maps that went through five one-product layers)
        return
    if hw[0] * hw[1] < 160 * 160:
        np.testing.assert_allclose(gx, gx_ref, rtol=0, atol=2e-5 * xr.grad.abs().max().item())
        flip = 1.0
    else:
        # 4.6 M ReLU / pooling decisions: one of them falls the other way in fp32 than in float64 (a pre-activation within rounding of zero), and its
        # receptive field in d_x (16 x 16 pixels x 32 channels = 1 % of the map) differs by its whole contribution -- in both forms of the device
        # run alike (whole network and layer by layer give the same 8,318 elements).  Held: 98 % of the elements at the bar, the rest within 2 %
        # of the largest gradient; parameter gradients (sums over all pixels): 99.5 % of a tensor's elements at ten times their bar, the rest within 2 %
        err = np.abs(gx - gx_ref)
        bar = 2e-5 * xr.grad.abs().max().item()
        assert (err > bar).mean() <= 0.02 and err.max() <= 2e-2 * xr.grad.abs().max().item(), ((err > bar).mean(), err.max())
        flip = 10.0
    for (name, pd), pr in zip(net_d.named_parameters(), ref.parameters()):
        got, want, top = pd.grad.cpu().numpy(), pr.grad.float().numpy(), pr.grad.abs().max().item()
        if flip == 1.0:
            np.testing.assert_allclose(got, want, rtol=0, atol=5e-5 * top, err_msg=name)
        else:                                            # (the flipped unit's own weight row carries its whole contribution: 0.12 % of down2's weight gradient)
            err = np.abs(got - want)
            assert (err > flip * 5e-5 * top).sum() <= max(2, 0.005 * err.size) and err.max() <= 2e-2 * top, (name, (err > flip * 5e-5 * top).sum(), err.max(), top)


@pytest.mark.parametrize("R,k,Cn,normalize", [(25600, 20, 3, True), (1000, 5, 3, False), (77, 1, 8, True)])
def test_composite_matches_torch_expression(R, k, Cn, normalize):
    """papr_composite_fwd / _bwd against the reference's line rgb = fg (1 - a) + bkg a (models/model.py:536-545) in torch ops on the
    CPU in float64: output and the gradients of fg, attn (zero outside the background column) and bkg_feats."""
    from papr_amd import ops
    gen = torch.Generator().manual_seed(R + k)
    fg = torch.rand(1, R, 1, Cn, generator=gen).requires_grad_(True)
    attn = torch.softmax(torch.randn(R, k + 1, generator=gen), -1).requires_grad_(True)
    bkg = torch.rand(1, Cn, generator=gen).requires_grad_(True)
    a = attn.double().reshape(1, R, 1, k + 1)[..., k:]
    ref = fg.double() * (1 - a) + bkg.double().reshape(1, 1, 1, -1) * a if normalize else fg.double() + bkg.double().reshape(1, 1, 1, -1) * a
    gy = torch.randn(ref.shape, generator=gen)
    (ref * gy.double()).sum().backward()
    d = dev()
    leaves = [t.detach().to(d).requires_grad_(True) for t in (fg, attn, bkg)]
    out = ops._CompositeFn.apply(leaves[0], leaves[1], leaves[2], normalize)
    (out * gy.to(d)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().float().numpy(), rtol=0, atol=3e-7)
    for got, want in zip(leaves, (fg, attn, bkg)):
        np.testing.assert_allclose(got.grad.cpu().numpy(), want.grad.numpy(), rtol=0, atol=2e-6 * (1 + want.grad.abs().max().item()))


def test_own_adam_step_matches_torch_adam():
    """papr_adam_step (one launch for all optimizers of PAPR.step, reference models/model.py:439-460) against torch.optim.Adam on
    the CPU in float64-free plain form: five steps, three optimizers with different learning rates / weight decay, a parameter
    without gradient, a channels-last parameter, sizes that are no multiple of 4; the state tensors are the torch optimizers' own
    (a state_dict round trip in the middle must not matter)."""
    from papr_amd import adam as own_adam
    gen = torch.Generator().manual_seed(11)
    shapes = [[(1000, 3)], [(256, 117), (256,), (7,), (32, 16, 3, 3)], [(5, 1)]]
    hyper = [dict(lr=2e-3, weight_decay=0.0), dict(lr=3e-4, weight_decay=1e-2), dict(lr=1e-3, weight_decay=0.0)]
    ref_params = [[torch.randn(*sh, generator=gen).requires_grad_(True) for sh in group] for group in shapes]
    ref_opts = [torch.optim.Adam(ps, **h) for ps, h in zip(ref_params, hyper)]
    d = dev()
    own_params = [[p.detach().clone().to(d).requires_grad_(True) for p in group] for group in ref_params]
    own_params[1][3] = own_params[1][3].detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    own_opts = [torch.optim.Adam(ps, **h) for ps, h in zip(own_params, hyper)]
    assert own_adam.supported(own_opts)
    for it in range(5):
        for gi, (rg, og) in enumerate(zip(ref_params, own_params)):
            for pi, (rp, op) in enumerate(zip(rg, og)):
                if gi == 1 and pi == 2 and it < 2:          # a parameter that gets no gradient at first
                    rp.grad = op.grad = None
                    continue
                g = torch.randn(rp.shape, generator=gen) * (10.0 ** -it)
                rp.grad = g.clone()
                op.grad = g.to(d)
        for o in ref_opts:
            o.step()
        own_adam.step(own_opts)
        if it == 2:                                        # checkpoint round trip of the torch optimizers (reference model.py:562-586)
            for o in own_opts:
                o.load_state_dict(o.state_dict())
        for o in ref_opts + own_opts:                       # a scheduler at work
            o.param_groups[0]["lr"] *= 0.9
    for rg, og, ro, oo in zip(ref_params, own_params, ref_opts, own_opts):
        for rp, op in zip(rg, og):
            np.testing.assert_allclose(op.detach().cpu().numpy(), rp.detach().numpy(), rtol=0, atol=2e-6 * (1 + rp.abs().max().item()))
            if rp in ro.state:
                assert float(oo.state[op]["step"]) == float(ro.state[rp]["step"])
                np.testing.assert_allclose(oo.state[op]["exp_avg"].cpu().numpy(), ro.state[rp]["exp_avg"].numpy(), rtol=1e-5, atol=1e-9)
                np.testing.assert_allclose(oo.state[op]["exp_avg_sq"].cpu().numpy(), ro.state[rp]["exp_avg_sq"].numpy(), rtol=1e-5, atol=1e-12)


def test_own_adam_under_a_gradscaler_follows_torchs_scaler_step():
    """papr_adam_step_scaled through papr_amd.adam.step_scaled (ABI 25; `use_amp: true`: reference models/model.py:439-442 steps every optimizer
    through `scaler.step`) against torch's own route -- torch.amp.GradScaler.step on fused torch.optim.Adam -- on the same scaled gradients: seven
    steps of three optimizers, an overflowing step in the middle (inf in one tensor of ONE optimizer: that optimizer stands still, the others step, the scale
    halves), then the scale grows again (growth_interval 2).  Parameters, moments, step counters and the scale agree step by step."""
    from papr_amd import adam as own_adam
    gen = torch.Generator().manual_seed(12)
    d = dev()
    shapes = [[(1000, 3)], [(256, 117), (256,), (7,)], [(5, 1)]]
    hyper = [dict(lr=2e-3, weight_decay=0.0), dict(lr=3e-4, weight_decay=1e-2), dict(lr=1e-3, weight_decay=0.0)]
    base = [[torch.randn(*sh, generator=gen) for sh in group] for group in shapes]
    ref_params = [[p.clone().to(d).requires_grad_(True) for p in group] for group in base]
    own_params = [[p.clone().to(d).requires_grad_(True) for p in group] for group in base]
    ref_opts = [torch.optim.Adam(ps, fused=True, **h) for ps, h in zip(ref_params, hyper)]
    own_opts = [torch.optim.Adam(ps, fused=True, **h) for ps, h in zip(own_params, hyper)]
    ref_sc = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=2)
    own_sc = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=2)
    assert own_adam.supported(own_opts)
    scales = []
    for it in range(7):
        # GradScaler.scale() initialises the scale tensor lazily: call it like train_step does
        ref_sc.scale(torch.zeros((), device=d)); own_sc.scale(torch.zeros((), device=d))
        sc = ref_sc.get_scale()
        assert own_sc.get_scale() == sc
        for gi, (rg, og) in enumerate(zip(ref_params, own_params)):
            for pi, (rp, op) in enumerate(zip(rg, og)):
                g = torch.randn(rp.shape, generator=gen) * (10.0 ** -(it % 3)) * sc
                if it == 3 and gi == 1 and pi == 1:
                    g[5] = float("inf")
                rp.grad = g.to(d)
                op.grad = g.to(d)
        before = [[p.detach().clone() for p in g_] for g_ in own_params]
        for o in ref_opts:
            ref_sc.step(o)
        own_adam.step_scaled(own_opts, own_sc)
        ref_sc.update(); own_sc.update()
        scales.append(own_sc.get_scale())
        assert own_sc.get_scale() == ref_sc.get_scale()
        if it == 3:                                   # (the optimizer that overflowed stands still; the other two step, as under scaler.step(opt) one by one)
            for b, p in zip(before[1], own_params[1]):
                assert torch.equal(b, p.detach()), "an overflowed optimizer moved a parameter"
            assert not torch.equal(before[0][0], own_params[0][0].detach())
        for rg, og, ro, oo in zip(ref_params, own_params, ref_opts, own_opts):
            for rp, op in zip(rg, og):
                np.testing.assert_allclose(op.detach().cpu().numpy(), rp.detach().cpu().numpy(), rtol=0, atol=2e-6 * (1 + rp.abs().max().item()))
                assert float(oo.state[op]["step"]) == float(ro.state[rp]["step"]), (it, float(oo.state[op]["step"]), float(ro.state[rp]["step"]))
                ma, va = ro.state[rp]["exp_avg"].cpu().numpy(), ro.state[rp]["exp_avg_sq"].cpu().numpy()
                np.testing.assert_allclose(oo.state[op]["exp_avg"].cpu().numpy(), ma, rtol=1e-5, atol=1e-6 * np.abs(ma).max())      # (moments near zero: cancellation)
                np.testing.assert_allclose(oo.state[op]["exp_avg_sq"].cpu().numpy(), va, rtol=1e-5, atol=1e-6 * np.abs(va).max())
    assert scales[3] == 0.5 * scales[2] and max(scales) >= 2048.0, scales      # halved by the overflow, grown every second clean step
    with pytest.raises(RuntimeError):               # two steps without an update(): the scaler's own error
        own_sc.scale(torch.zeros((), device=d))
        for p in own_params[0]:
            p.grad = torch.zeros_like(p)
        own_adam.step_scaled(own_opts, own_sc)
        own_adam.step_scaled(own_opts, own_sc)


@pytest.mark.parametrize("mode", ["f32", "fwd", "dgrad", "layers"])
def test_other_gemm_modes_meet_the_same_mlp_parity(mode):
    """PAPR_GEMM_MODE selects which wide GEMMs use the split-f16 (hi/lo, 3 MFMA) kernels (default `h3`: forward
    layers, data-gradients and weight-gradients); `dgrad` (no weight-gradients), `fwd` (forward only) and `f32`
    (fp32 MFMA everywhere) must pass the same tests.  The switch is read when the library loads, hence the
    child process."""
    import os, subprocess, sys
    env = dict(os.environ, PAPR_GEMM_MODE=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "mlp_forward_backward"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:]


@pytest.mark.parametrize("act,skips,hw", [("leakyrelu", [], (20, 13)), ("relu", [1], (8, 8))])
@pytest.mark.parametrize("rows", ["f32", "f16"])
def test_mlp_generator_matches_torch(act, skips, hw, rows, monkeypatch):
    """MLPGenerator (per-pixel MLP render head, reference models/renderer.py:6-17) on the HIP kernels against the same
    module written with torch ops: forward and all gradients."""
    from papr_amd.unet import MLPGenerator
    monkeypatch.setenv("PAPR_H3_ROWS", rows)               # (f16: the default -- weight gradients from f16-rounded rows, see test_mlp_forward_backward_vs_torch)
    torch.manual_seed(5)
    gen = MLPGenerator(32, 3, 128, 3, act_type=act, last_act_type="none", skip_layers=skips)
    x = torch.randn(2, 32, *hw, requires_grad=True)
    lin = gen.mlp.linears()
    h = inp = x.permute(0, 2, 3, 1)
    for i, m in enumerate(lin):
        if i in skips:
            h = torch.cat([h, inp], -1)
        h = torch.nn.functional.linear(h, m.weight, m.bias)
        if i < len(lin) - 1:
            h = torch.relu(h) if act == "relu" else torch.nn.functional.leaky_relu(h, 0.2)
    y_ref = h.permute(0, 3, 1, 2)
    w = torch.randn_like(y_ref)
    (y_ref * w).sum().backward()
    ref_grads = [p.grad.clone() for p in gen.parameters()] + [x.grad.clone()]
    for p in gen.parameters():
        p.grad = None
    d = dev()
    gen_d = gen.to(d)
    xd = x.detach().to(d).requires_grad_(True)
    y = gen_d(xd)
    (y * w.to(d)).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), rtol=0, atol=2e-5 * max(1.0, y_ref.abs().max().item()))
    got = [p.grad.cpu() for p in gen_d.parameters()] + [xd.grad.cpu()]
    for j, (g, r) in enumerate(zip(got, ref_grads)):
        wtol = 3e-5 if (rows == "f32" or j == len(got) - 1) else 2e-3      # (the input gradient -- last entry -- is the same computation either way)
        np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=0, atol=wtol * (r.abs().max().item() + 1e-12))
    with torch.no_grad():                                   # inference route (nothing saved) gives the same numbers
        assert torch.equal(gen_d(xd.detach()), y.detach())


@pytest.mark.parametrize("P,k", [(10000, 10), (3001, 4), (17, 16), (5, 5)])
def test_points_knn_matches_kdtree(P, k):
    """papr_points_knn (cloud.hip) against scipy's KDTree -- what add_points_knn queries (models/utils.py:27-29, 59): the same
    neighbours in the same order and the same double-precision distances, for every point of the cloud and for a subset."""
    from scipy.spatial import KDTree
    from papr_amd import ops
    gen = torch.Generator().manual_seed(P + k)
    pts = (torch.rand(P, 3, generator=gen) * 2 - 1).float()
    tree = KDTree(pts.numpy())
    d, i = tree.query(pts.numpy(), k=k)
    d, i = d.reshape(P, k), i.reshape(P, k)
    nn_i, nn_d = ops.points_knn(pts.to("cuda:0"), k)
    torch.cuda.synchronize()
    assert np.array_equal(nn_i.cpu().numpy(), i)
    np.testing.assert_allclose(nn_d.cpu().numpy(), d, rtol=1e-15, atol=0)
    q = torch.randperm(P, generator=gen)[: max(1, P // 3)].int()
    nn_i, nn_d = ops.points_knn(pts.to("cuda:0"), k, query_idx=q.to("cuda:0"))
    torch.cuda.synchronize()
    assert np.array_equal(nn_i.cpu().numpy(), i[q.long().numpy()])
    np.testing.assert_allclose(nn_d.cpu().numpy(), d[q.long().numpy()], rtol=1e-15, atol=0)


@pytest.mark.parametrize("comb,sample", [("random", "top-knn-std"), ("mean", "top-knn-max"), ("weighted", "random"), ("random-softmax", "influ-scores-max"),
                                         ("duplicate", "top-knn-mean")])
def test_grow_points_on_the_device_equals_the_host_procedure(comb, sample):
    """pointcloud.grow_points_device (the cloud stays on the GPU; neighbour searches, ranking and blends on the device) against
    pointcloud.grow_points (the reference's host procedure, models/utils.py:9-109, pinned by tests/test_host_model.py) under the
    same numpy seed: the same sites, the same new points / influence scores / features."""
    from papr_amd.pointcloud import grow_points, grow_points_device
    gen = torch.Generator().manual_seed(11)
    P, add, k = 4000, 300, 3
    pts = (torch.rand(P, 3, generator=gen) * 2 - 1).float()
    influ = torch.randn(P, 1, generator=gen)
    feats = torch.randn(P, 64, generator=gen)
    np.random.seed(5)
    want = grow_points(pts, influ, add, k, comb_type=comb, sample_type=sample, sample_k=10, feats=feats)
    np.random.seed(5)
    got = grow_points_device(pts.to("cuda:0"), influ.to("cuda:0"), add, k, comb_type=comb, sample_type=sample, sample_k=10, feats=feats.to("cuda:0"))
    torch.cuda.synchronize()
    assert got[1] == want[1] == add
    for g, w, name in ((got[0], want[0], "coords"), (got[2], want[2], "influ"), (got[3], want[3], "feats")):
        np.testing.assert_allclose(g.cpu().numpy(), np.asarray(w), rtol=0, atol=2e-6, err_msg=name)


@pytest.mark.parametrize("R,dq,dm,ldq", [(25600, 64, 256, 64), (777, 48, 96, 64), (3, 5, 7, 8), (300, 1024, 40, 1024)])
def test_qk_bias_bwd_rank_one_terms_equal_the_torch_formulas(R, dq, dm, ldq):
    """papr_qk_bias_bwd (the gradient of the score bias c0 = q'.b_k, q' = W_q Q + b_q; models/attn.py:217-225) against autograd of that very
    expression in float64."""
    from papr_amd import hip
    lib = hip.lib()
    g = torch.Generator().manual_seed(R + dq)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)
    Q, d_c0, wq, bk, bq = rnd(R, ldq), rnd(R) * 0.01, rnd(dm, ldq), rnd(dm), rnd(dm)
    d_Q0, d_wq0, d_bq0 = rnd(R, ldq), rnd(dm, ldq), rnd(dm)
    dev = "cuda"
    c = lambda t: t.to(dev).contiguous()
    d_Q, d_wq, d_bq, d_bk = c(d_Q0), c(d_wq0), c(d_bq0), torch.empty(dm, device=dev)
    ws = torch.empty((lib.papr_qk_bias_bwd_workspace_bytes(dq) + 3) // 4, device=dev)
    Qd, gd, wqd, bkd, bqd = c(Q), c(d_c0), c(wq), c(bk), c(bq)
    hip.check(lib.papr_qk_bias_bwd(hip.ptr(Qd), ldq, dq, dm, hip.ptr(gd), R, hip.ptr(wqd), ldq, hip.ptr(bkd), hip.ptr(bqd), hip.ptr(d_Q), hip.ptr(d_wq),
                                   hip.ptr(d_bq), hip.ptr(d_bq), hip.ptr(d_bk), hip.ptr(ws), hip.stream_ptr()), "papr_qk_bias_bwd")
    A = lambda t: t.double().clone().requires_grad_(True)
    Qa, wqa, bka, bqa = A(Q[:, :dq]), A(wq[:, :dq]), A(bk), A(bq)
    c0 = (Qa @ wqa.t() + bqa) @ bka
    c0.backward(d_c0.double())
    e_Q, e_wq = d_Q0.double().clone(), d_wq0.double().clone()
    e_Q[:, :dq] += Qa.grad
    e_wq[:, :dq] += wqa.grad
    for name, got, exp in (("d_bk", d_bk, bka.grad), ("d_Q", d_Q, e_Q), ("d_wq", d_wq, e_wq), ("d_bq", d_bq, d_bq0.double() + bqa.grad)):
        err = (got.cpu().double() - exp).abs().max().item()
        assert err <= 2e-6 * max(exp.abs().max().item(), 1.0) * max(1.0, (R / 1000.0) ** 0.5), (name, err)
    # the padding columns are untouched
    assert torch.equal(d_Q.cpu()[:, dq:], d_Q0[:, dq:]) and torch.equal(d_wq.cpu()[:, dq:], d_wq0[:, dq:])


@pytest.mark.parametrize("shape", [(1, 160, 160, 3), (7,), (3, 5, 11), (1, 1)])
def test_own_mse_loss_equals_torch_mse_loss_and_its_gradient(shape):
    """papr_amd.loss.MSELoss on the device (papr_mse_fwd: loss and gradient direction in one launch) against torch.nn.MSELoss in float64."""
    from papr_amd.loss import MSELoss, _MseFn
    g = torch.Generator().manual_seed(len(shape))
    pred, tgt = torch.rand(*shape, generator=g), torch.rand(*shape, generator=g)
    pr = pred.double().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(pr, tgt.double())
    (ref * 3.0).backward()
    pd = pred.to(dev()).requires_grad_(True)
    got = MSELoss()(pd, tgt.to(dev()))
    assert isinstance(got.grad_fn, _MseFn._backward_cls)
    (got * 3.0).backward()
    assert abs(got.item() - ref.item()) <= 2e-7 * ref.item()
    np.testing.assert_allclose(pd.grad.cpu().numpy(), pr.grad.float().numpy(), rtol=2e-6, atol=0)


@pytest.mark.parametrize("c_in,n_classes,hw,batch", [(64, 4, (16, 24), 1), (32, 1, (8, 8), 3), (96, 2, (4, 12), 1)])
def test_small_unet_whole_network_call_other_channel_counts(c_in, n_classes, hw, batch):
    """papr_small_unet_fwd / _bwd beyond the shipped 32 -> 3 head: other input widths (multiples of 32), 1 .. 4 classes, the smallest maps (4 pixels
    on a side: 1 x 1 at the bottom of the U), batches -- output and every gradient against the same module in float64 on the CPU."""
    import copy
    from papr_amd.unet import SmallUNet
    torch.manual_seed(c_in + n_classes)
    net = SmallUNet(c_in, n_classes)
    x = torch.randn(batch, c_in, *hw)
    gy = torch.randn(batch, n_classes, *hw) * 1e-2
    ref = copy.deepcopy(net).double()
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    (yr * gy.double()).sum().backward()
    d = dev()
    net_d = net.to(d)
    xd = x.to(d).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        yd = net_d(xd)
        (yd * gy.to(d)).sum().backward()
        torch.cuda.synchronize()
    assert any("unet_prep" in e.key for e in prof.key_averages())              # (the whole-network path ran)
    np.testing.assert_allclose(yd.detach().cpu().numpy(), yr.detach().float().numpy(), rtol=0, atol=1e-5 * yr.abs().max().item())
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.float().numpy(), rtol=0, atol=2e-5 * xr.grad.abs().max().item())
    for (name, pd), pr in zip(net_d.named_parameters(), ref.parameters()):
        np.testing.assert_allclose(pd.grad.cpu().numpy(), pr.grad.float().numpy(), rtol=0, atol=5e-5 * pr.grad.abs().max().item(), err_msg=name)
    # no_grad (nothing kept for a backward pass) gives the same bits
    with torch.no_grad():
        assert torch.equal(net_d(xd), yd)
