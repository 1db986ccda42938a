"""Host-side point-cloud growth (runs every `add_steps` steps, between training steps).

Counterpart of the reference's add_points_knn (models/utils.py:9-109): choose the sparsest regions
of the cloud by the spread of k-nearest-neighbour distances (scipy KDTree on the CPU, exactly as the
reference does) and place one new point per chosen site as a random convex combination of its
neighbours.  Not on the per-ray hot path.
"""
import numpy as np
import scipy.special
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
