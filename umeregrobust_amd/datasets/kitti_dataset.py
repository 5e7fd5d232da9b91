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
    """reference utils/general_utils.py:16-19.  The cache format IS a pickle (kitti_dataset.py:647-657), and unpickling
    executes what the file says: read cache files only from a source you trust (your own preprocessing run)."""
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


class _Dilution:
    """One cloud of one batch element thinned to `size` points: the draw (`rng.choice(n, size, replace=False)`, the call the
    reference makes at kitti_dataset.py:571 / :579) and the inverse map original index -> position in the thinned cloud
    (-1: dropped), which is what the match bookkeeping below needs."""

    def __init__(self, n, size, rng):
        self.keep = rng.choice(n, size, replace=False)
        self.position = np.full(n, -1, dtype=np.int64)
        self.position[self.keep] = np.arange(size, dtype=np.int64)

    def __call__(self, *fields):
        return tuple(f[self.keep] for f in fields)


def surviving_matches(matches, src, tgt):
    """The correspondences (source index, target index) whose two end points both survive the dilutions `src` / `tgt`,
    re-indexed into the thinned clouds -- the result of the reference's two `np.intersect1d(..., return_indices=True)` calls
    (kitti_dataset.py:585-589), expressed through the inverse maps:
      * a source point keeps only its first correspondence (in file order); candidates are visited in ascending source index;
      * a target point keeps only the first of those candidates; rows come out in ascending target index."""
    matches = np.asarray(matches)
    _, first_of_src = np.unique(matches[:, 0], return_index=True)              # ascending source index, first occurrence
    rows = first_of_src[src.position[matches[first_of_src, 0]] >= 0]
    _, first_of_tgt = np.unique(matches[rows, 1], return_index=True)           # ascending target index, first candidate
    rows = rows[first_of_tgt]
    rows = rows[tgt.position[matches[rows, 1]] >= 0]
    return np.stack([src.position[matches[rows, 0]], tgt.position[matches[rows, 1]]], axis=1)


def batch_collate_fn_dset(data, num_matches, max_pc_size=100000, rng=np.random):
    """The loader output contract of reference datasets/kitti/kitti_dataset.py:546-616 (what evaluate.py:175-178 unpacks).
    data: list of items (src_pts, src_sem, src_coords, tgt_pts, tgt_sem, tgt_coords, src_pts_tform, gt_tform, matches
    [, src_feat, tgt_feat]).  Every cloud of the batch is thinned to the batch-minimum size (at most max_pc_size).  The
    host RNG is consumed in the reference's order, on the reference's (global numpy) stream by default: per element the
    source draw, then the target draw; after the loop one draw per element for the matches.
    -> (src_pts [bs,n,3], src_seg [bs,n], src_coords [bs*n,4], src_feat [bs*n,1] (ones: the network's input),
        tgt_pts, tgt_seg, tgt_coords, tgt_feat, src_pts_tform [bs,n,3], gt_tform [bs,4,4], matches [bs,k,2])
       + (src_net_feat [bs,n,d], tgt_net_feat [bs,m,d]) when the items carry features."""
    with_feat = len(data[0]) > 9
    n_src = min(min(len(d[0]) for d in data), max_pc_size)                     # :565-566
    n_tgt = min(min(len(d[3]) for d in data), max_pc_size)
    sides = {"src": [], "tgt": []}                                             # per element: (pts, seg, coords, ones[, net feature])
    moved, kept_matches = [], []
    for d in data:
        src = _Dilution(len(d[0]), n_src, rng)                                 # :571
        tgt = _Dilution(len(d[3]), n_tgt, rng)                                 # :579
        for name, dil, first, feat_at in (("src", src, 0, 9), ("tgt", tgt, 3, 10)):
            pts, seg, coords = dil(d[first], d[first + 1], d[first + 2])
            fields = (pts, seg, coords, torch.ones_like(pts[:, :1]).float())
            sides[name].append(fields + (dil(d[feat_at]) if with_feat else ()))
        moved.append(src(d[6])[0])
        kept_matches.append(surviving_matches(d[8], src, tgt))                 # :585-589

    def batched(name):
        pts, seg, coords, ones = (list(col) for col in list(zip(*sides[name]))[:4])
        coords, ones = sparse_collate(coords, ones)                            # :593, :599
        return torch.stack(pts, dim=0), torch.stack(seg, dim=0), coords, ones

    k = min(min(len(m) for m in kept_matches), num_matches)                    # :606-607
    matches = torch.stack([torch.from_numpy(m[rng.choice(len(m), k, replace=False)]) for m in kept_matches], dim=0)
    out = batched("src") + batched("tgt") + (torch.stack(moved, dim=0), torch.stack([d[7] for d in data], dim=0), matches)
    if with_feat:
        out = out + tuple(torch.stack([e[4] for e in sides[name]], dim=0) for name in ("src", "tgt"))
    return out


def checkpoint_state_dict(path_or_obj, allow_pickle=False):
    """The weight-file schema of reference train_coloring.py:214-222 as `evaluate.py:164` consumes it: a dict with
    'epoch', 'model_state_dict', 'optimizer_state_dict', 'total_loss'.  -> the model state dict (name -> tensor); raises
    with the offending keys otherwise.  (Bare state dicts, `save_model` :209-211, are accepted as they are.)
    Files are read with torch's restricted unpickler (`weights_only=True`); a checkpoint that needs the full one (arbitrary
    pickled objects = arbitrary code at load time) is refused unless `allow_pickle=True`: only for files you trust."""
    if isinstance(path_or_obj, (str, os.PathLike)):
        try:
            # tensors, containers and scalars only: nothing in the file is executed
            ck = torch.load(path_or_obj, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError:
            if not allow_pickle:
                raise
            ck = torch.load(path_or_obj, map_location="cpu", weights_only=False)     # full unpickler: trusted files only
    else:
        ck = path_or_obj
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
