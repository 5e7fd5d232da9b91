"""SURVEY 8(f4), the data formats in front of the hot path -- what can be built without the datasets, the MinkUNet
weights and a sparse-convolution backend:

  * the reference's pre-processed pair cache, one pickle per pair
    `<cache_data_path>/<split>/<seq:02d>/<frame0:06d>_<frame1:06d>.pickle` with the nine keys written at
    reference datasets/kitti/kitti_dataset.py:647-655 and read back by `cached_getitem` (:441-458);
  * `batch_collate_fn_dset` (:546-616), the loader output contract `evaluate.py:175-178` unpacks: the per-pair random
    dilution to `max_pc_size` points with the HOST numpy RNG (the draws come in the reference's order, so a seeded run
    consumes the same stream), the matches that survive the dilution, MinkowskiEngine's `sparse_collate` layout
    (batch index in column 0 of the coordinates);
  * the weight-file schema of reference train_coloring.py:214-222 (`evaluate.py:164` reads `['model_state_dict']`).

The feature network itself (reference models.py:691-698) needs MinkowskiEngine: not installable here, so features stay
an input.  A cache pickle may carry two extra keys, `src_feat` / `tgt_feat` [n,32] (the network's output at the cached
points, dumped by whoever can run it); the collate function then dilutes them with the same indices and appends them
to its return tuple, and `python -m umeregrobust_amd.evaluate --cache <dir>` runs the reference's loop from such a cache.

MinkowskiEngine 0.5.4 `sparse_collate` is restated from its documented behaviour (parity unpinned, like the other
third-party boundaries); everything else is pinned by golden G10 (the reference's own collate on seeded items)."""
import glob
import os
import pickle

import numpy as np
import torch

CACHE_KEYS = ("src_pts", "src_seg", "src_coords", "tgt_pts", "tgt_seg", "tgt_coords", "src_pts_tform", "gt_tform", "matches")
FEATURE_KEYS = ("src_feat", "tgt_feat")


def load_pickle(filename):
    """reference utils/general_utils.py:16-19"""
    with open(filename, "rb") as f:
        return pickle.load(f)


def read_cached_pair(path, with_features=False):
    """One cache file -> the 9-tuple of `cached_getitem` (reference kitti_dataset.py:441-458):
    (src_pts f32[n,3], src_seg i64[n], src_coords i32[n,3], tgt_pts, tgt_seg, tgt_coords, src_pts_tform f32[n,3],
    gt_tform f32[4,4], matches i64[m,2]); with_features: + (src_feat f32[n,d], tgt_feat f32[m,d]) if the file has them."""
    d = load_pickle(path)
    missing = [k for k in CACHE_KEYS if k not in d]
    if missing:
        raise KeyError(f"{path}: not a pair cache file, keys {missing} are missing (expected {list(CACHE_KEYS)})")
    item = tuple(d[k] for k in CACHE_KEYS)
    if with_features:
        if not all(k in d for k in FEATURE_KEYS):
            raise KeyError(f"{path}: no `src_feat` / `tgt_feat`: the feature network (reference models.py:691-698, "
                           "MinkowskiEngine) is not part of this library -- dump its outputs into the cache files")
        item = item + tuple(d[k] for k in FEATURE_KEYS)
    return item


def write_cached_pair(path, item, src_feat=None, tgt_feat=None):
    """The writer of reference kitti_dataset.py:647-657 (same keys, same pickle protocol), plus the optional features."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    d = dict(zip(CACHE_KEYS, item[:9]))
    if src_feat is not None:
        d["src_feat"], d["tgt_feat"] = src_feat, tgt_feat
    with open(path, "wb") as handle:
        pickle.dump(d, handle, protocol=pickle.HIGHEST_PROTOCOL)


class CachedPairDataset(torch.utils.data.Dataset):
    """The cache-backed half of the reference's SemanticKITTIDataset / NuscenesDataset (`cache_data_path != ""`,
    kitti_dataset.py:380-385): item i = the pickle of pair i.  The reference takes its pair list from the raw dataset's
    pose files; here it is the sorted content of `<cache_data_path>/<split>/*/` (or an explicit list of
    (seq_id, frame0_id, frame1_id))."""

    def __init__(self, cache_data_path, split="test", files=None, with_features=False, dataset="kitti"):
        """dataset: "kitti" -- sequence ids are integers, directories `%02d` (kitti_dataset.py:444); "nuscenes" -- sequence
        ids are the directory names themselves (nuscenes_dataset.py:452)."""
        self.cache_data_path, self.split, self.with_features, self.dataset = cache_data_path, split, with_features, dataset
        if files is None:
            files = []
            for p in sorted(glob.glob(os.path.join(cache_data_path, split, "*", "*.pickle"))):
                seq, name = os.path.basename(os.path.dirname(p)), os.path.splitext(os.path.basename(p))[0]
                f0, f1 = name.split("_")
                files.append((int(seq) if dataset == "kitti" else seq, int(f0), int(f1)))
        self.files = list(files)

    def path(self, idx):
        seq_id, frame0_id, frame1_id = self.files[idx]
        seq_dir = seq_id if isinstance(seq_id, str) else f"{seq_id:02d}"
        return os.path.join(self.cache_data_path, self.split, seq_dir, f"{frame0_id:06d}_{frame1_id:06d}.pickle")

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        return read_cached_pair(self.path(idx), self.with_features)


def sparse_collate(coords, feats):
    """MinkowskiEngine.utils.sparse_collate as called at reference kitti_dataset.py:593,599 (no labels): coordinates
    [n_i, D] of every batch element stacked with the batch index in a new column 0 (int32), features concatenated."""
    bcoords, bfeats = [], []
    for b, (c, f) in enumerate(zip(coords, feats)):
        c = torch.as_tensor(c)
        c = (torch.floor(c) if c.is_floating_point() else c).to(torch.int32)
        bcoords.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32), c], dim=1))
        bfeats.append(torch.as_tensor(f))
    return torch.cat(bcoords, dim=0), torch.cat(bfeats, dim=0)


def batch_collate_fn_dset(data, num_matches, max_pc_size=100000, rng=np.random):
    """reference datasets/kitti/kitti_dataset.py:546-616.
    data: list of items (src_pts, src_sem, src_coords, tgt_pts, tgt_sem, tgt_coords, src_pts_tform, gt_tform, matches
    [, src_feat, tgt_feat]).  Every cloud of the batch is diluted to the batch-minimum size (at most max_pc_size) by
    `rng.choice(n, size, replace=False)` -- source then target, element by element, then one draw per element for the
    matches: the reference's order on the reference's (global numpy) stream.
    -> (src_pts [bs,n,3], src_seg [bs,n], src_coords [bs*n,4], src_feat [bs*n,1] (ones: the network's input),
        tgt_pts, tgt_seg, tgt_coords, tgt_feat, src_pts_tform [bs,n,3], gt_tform [bs,4,4], matches [bs,k,2])
       + (src_net_feat [bs,n,d], tgt_net_feat [bs,m,d]) when the items carry features."""
    bs = len(data)
    with_feat = len(data[0]) > 9
    src_pts, src_seg, src_coords, src_feat, tgt_pts, tgt_seg, tgt_coords, tgt_feat = [], [], [], [], [], [], [], []
    src_pts_tform, matches, src_net, tgt_net = [], [], [], []
    src_num_pts = min(min(len(d[0]) for d in data), max_pc_size)               # :565-566
    tgt_num_pts = min(min(len(d[3]) for d in data), max_pc_size)
    for b_idx in range(bs):
        d = data[b_idx]
        src_rand_idx = rng.choice(len(d[0]), src_num_pts, replace=False)        # :571
        src_pts.append(d[0][src_rand_idx])
        src_seg.append(d[1][src_rand_idx])
        src_coords.append(d[2][src_rand_idx])
        src_feat.append(torch.ones_like(d[0][src_rand_idx, :1]).float())
        src_pts_tform.append(d[6][src_rand_idx])
        tgt_rand_idx = rng.choice(len(d[3]), tgt_num_pts, replace=False)        # :579
        tgt_pts.append(d[3][tgt_rand_idx])
        tgt_seg.append(d[4][tgt_rand_idx])
        tgt_coords.append(d[5][tgt_rand_idx])
        tgt_feat.append(torch.ones_like(d[3][tgt_rand_idx, :1]).float())
        if with_feat:
            src_net.append(d[9][src_rand_idx])
            tgt_net.append(d[10][tgt_rand_idx])
        # matches that survive the dilution, re-indexed into the diluted clouds (:585-589)
        m = np.asarray(d[8])
        _, m1, idxs1 = np.intersect1d(src_rand_idx, m[:, 0], return_indices=True)
        _, m2, idxs2 = np.intersect1d(tgt_rand_idx, m[idxs1, 1], return_indices=True)
        matches.append(np.concatenate([m1[idxs2, None], m2[:, None]], axis=1))
    src_coords, src_feat = sparse_collate(src_coords, src_feat)                 # :593
    tgt_coords, tgt_feat = sparse_collate(tgt_coords, tgt_feat)                 # :599
    gt_tform = torch.stack([d[7] for d in data], dim=0)
    num_matches = min(min(len(m) for m in matches), num_matches)                 # :606-607
    matches = torch.stack([torch.from_numpy(m[rng.choice(len(m), num_matches, replace=False)]) for m in matches], dim=0)
    out = (torch.stack(src_pts, dim=0), torch.stack(src_seg, dim=0), src_coords, src_feat,
           torch.stack(tgt_pts, dim=0), torch.stack(tgt_seg, dim=0), tgt_coords, tgt_feat,
           torch.stack(src_pts_tform, dim=0), gt_tform, matches)
    if with_feat:
        out = out + (torch.stack(src_net, dim=0), torch.stack(tgt_net, dim=0))
    return out


def checkpoint_state_dict(path_or_obj):
    """The weight-file schema of reference train_coloring.py:214-222 as `evaluate.py:164` consumes it: a dict with
    'epoch', 'model_state_dict', 'optimizer_state_dict', 'total_loss'.  -> the model state dict (name -> tensor); raises
    with the offending keys otherwise.  (Bare state dicts, `save_model` :209-211, are accepted as they are.)"""
    ck = torch.load(path_or_obj, map_location="cpu", weights_only=False) if isinstance(path_or_obj, (str, os.PathLike)) else path_or_obj
    if not isinstance(ck, dict):
        raise TypeError(f"checkpoint: expected a dict, got {type(ck).__name__}")
    if "model_state_dict" in ck:
        missing = [k for k in ("epoch", "optimizer_state_dict", "total_loss") if k not in ck]
        if missing:
            raise KeyError(f"checkpoint: keys {missing} of the save_checkpoint schema are missing")
        sd = ck["model_state_dict"]
    else:
        sd = ck
    bad = [k for k, v in sd.items() if not isinstance(v, torch.Tensor)]
    if bad or not sd:
        raise TypeError(f"checkpoint: model_state_dict must map parameter names to tensors (offending: {bad[:5]})")
    return sd
