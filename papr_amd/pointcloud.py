"""Point-cloud growth (runs every `add_steps` steps, between training steps).

Counterpart of the reference's add_points_knn (models/utils.py:9-109): choose the sparsest regions
of the cloud by the spread of k-nearest-neighbour distances and place one new point per chosen site
as a random convex combination of its neighbours.  Not on the per-ray hot path (SURVEY.md section
8f, rank 3).

grow_points_device: the product path for a cloud on the GPU -- both neighbour searches, the ranking
and the blends run on the device (papr_points_knn, cloud.hip); only the numpy draws of the
reference (site choice / blend weights, the global numpy stream) are made on the host and uploaded,
so that one seed gives the reference's points.  grow_points: the reference's own host procedure
(scipy KDTree on CPU tensors), which a model living on the CPU uses.
"""
import numpy as np
import scipy.special
import torch
from scipy.spatial import KDTree


def _site_ranking(dists, mode):
    if mode == "top-knn-std":
        return dists.std(axis=-1)
    if mode == "top-knn-mean":
        return dists.mean(axis=-1)
    if mode == "top-knn-max":
        return dists.max(axis=-1)
    if mode == "top-knn-min":
        return dists.min(axis=-1)
    raise NotImplementedError(mode)


def grow_points(coords, influ, add_num, k, comb_type="mean", sample_type="random", sample_k=10, feats=None):
    """coords (P,3), influ (P,1), feats (P,F) CPU tensors -> (new_coords, n_new, new_influ, new_feats)."""
    tree = KDTree(coords)
    P = coords.shape[0]
    if P <= add_num and "random" in comb_type:
        sites = np.random.choice(P, add_num, replace=True)
    elif P <= add_num:
        sites = list(range(P))
    elif sample_type == "random":
        sites = np.random.choice(P, add_num, replace=False)
    elif sample_type.startswith("top-knn-"):
        assert k >= 2
        nn_d, _ = tree.query(coords, k=sample_k)
        sites = np.argsort(_site_ranking(nn_d, sample_type))[-add_num:]
    elif sample_type == "influ-scores-max":
        sites = np.argsort(influ.squeeze())[-add_num:]
    elif sample_type == "influ-scores-min":
        sites = np.argsort(influ.squeeze())[:add_num]
    else:
        raise NotImplementedError(sample_type)
    query = coords[sites, :]

    new_feats = None
    if comb_type == "duplicate":
        shift = np.random.randn(3).astype(np.float32)
        shift = shift / np.linalg.norm(shift) * k
        new_coords = query + shift
        new_influ = influ[sites, :]
        if feats is not None:
            new_feats = feats[sites, :]
        return new_coords, len(new_coords), new_influ, new_feats

    nn_d, nn_i = tree.query(query, k=k + 1)
    nn_d = nn_d.astype(np.float32)[:, 1:]
    nn_i = nn_i[:, 1:]
    if comb_type == "mean":
        w = np.full((query.shape[0], k), 1.0 / k, dtype=np.float32)
    elif comb_type == "random":
        w = np.random.uniform(0, 1, (query.shape[0], k)).astype(np.float32)
        w /= w.sum(axis=-1, keepdims=True)
    elif comb_type == "random-softmax":
        w = scipy.special.softmax(np.random.randn(query.shape[0], k).astype(np.float32), axis=-1)
    elif comb_type == "weighted":
        w = 1.0 / (nn_d + 1e-6)
        w = w / w.sum(axis=-1, keepdims=True)
    else:
        raise NotImplementedError(comb_type)
    w3 = w.reshape(-1, k, 1)
    blend = (lambda t: t[nn_i, :].mean(axis=-2)) if comb_type == "mean" else (lambda t: (t[nn_i, :] * w3).sum(axis=-2))
    new_coords = blend(coords)
    new_influ = blend(influ)
    if feats is not None:
        new_feats = blend(feats)
    return new_coords, len(new_coords), new_influ, new_feats


def grow_points_device(coords, influ, add_num, k, comb_type="mean", sample_type="random", sample_k=10, feats=None):
    """coords (P,3), influ (P,1), feats (P,F) device tensors -> (new_coords, n_new, new_influ, new_feats), device tensors.

    Same procedure and the same numpy draws as grow_points; the neighbour distances are the KDTree's doubles.  The P ranking values
    (one double per point, not the cloud) go to the host for numpy's own argsort: exactly equal values are common (two points that
    are each other's farthest neighbour share their top-knn-max) and numpy orders them its own way."""
    from . import ops
    P, dev = coords.shape[0], coords.device
    coords = coords.detach().float().contiguous()
    if P <= add_num and "random" in comb_type:
        sites = torch.from_numpy(np.random.choice(P, add_num, replace=True)).to(dev)
    elif P <= add_num:
        sites = torch.arange(P, device=dev)
    elif sample_type == "random":
        sites = torch.from_numpy(np.random.choice(P, add_num, replace=False)).to(dev)
    elif sample_type.startswith("top-knn-"):
        assert k >= 2
        _, nn_d = ops.points_knn(coords, sample_k)
        mode = sample_type[len("top-knn-"):]
        if mode == "std":
            score = nn_d.std(dim=-1, unbiased=False)
        elif mode == "mean":
            score = nn_d.mean(dim=-1)
        elif mode == "max":
            score = nn_d.max(dim=-1).values
        elif mode == "min":
            score = nn_d.min(dim=-1).values
        else:
            raise NotImplementedError(sample_type)
        sites = torch.from_numpy(np.argsort(score.cpu().numpy())[-add_num:]).to(dev)
    elif sample_type == "influ-scores-max":
        sites = torch.from_numpy(np.argsort(influ.detach().reshape(-1).cpu().numpy())[-add_num:]).to(dev)
    elif sample_type == "influ-scores-min":
        sites = torch.from_numpy(np.argsort(influ.detach().reshape(-1).cpu().numpy())[:add_num]).to(dev)
    else:
        raise NotImplementedError(sample_type)
    sites = sites.long()
    query = coords[sites, :]
    influ = influ.detach()
    feats = None if feats is None else feats.detach()

    if comb_type == "duplicate":
        shift = np.random.randn(3).astype(np.float32)
        shift = shift / np.linalg.norm(shift) * k
        return query + torch.from_numpy(shift).to(dev), query.shape[0], influ[sites, :], None if feats is None else feats[sites, :]

    nn_i, nn_d = ops.points_knn(coords, k + 1, query_idx=sites.int().contiguous())
    nn_i = nn_i[:, 1:].long()
    nn_d = nn_d[:, 1:].float()
    n = query.shape[0]
    if comb_type == "mean":
        w = None
    elif comb_type == "random":
        w = np.random.uniform(0, 1, (n, k)).astype(np.float32)
        w = torch.from_numpy(w / w.sum(axis=-1, keepdims=True)).to(dev)
    elif comb_type == "random-softmax":
        w = torch.from_numpy(scipy.special.softmax(np.random.randn(n, k).astype(np.float32), axis=-1)).to(dev)
    elif comb_type == "weighted":
        w = 1.0 / (nn_d + 1e-6)
        w = w / w.sum(dim=-1, keepdim=True)
    else:
        raise NotImplementedError(comb_type)
    blend = (lambda t: t[nn_i, :].mean(dim=-2)) if w is None else (lambda t: (t[nn_i, :] * w.reshape(-1, k, 1)).sum(dim=-2))
    return blend(coords), n, blend(influ), None if feats is None else blend(feats)
