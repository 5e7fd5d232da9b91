"""CPU: SURVEY 8(f4) -- the pair cache format, the loader output contract (golden G10 = the reference's own
batch_collate_fn_dset on seeded items) and the weight-file schema."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("src_pts", "src_seg", "src_coords", "tgt_pts", "tgt_seg", "tgt_coords", "src_pts_tform", "gt_tform", "matches")
OUT = ("src_pts", "src_seg", "src_coords", "src_feat", "tgt_pts", "tgt_seg", "tgt_coords", "tgt_feat", "src_pts_tform", "gt_tform", "matches")


def _items(g):
    return [tuple(torch.from_numpy(g[f"in{i}_{k}"]) for k in KEYS) for i in range(3)]


def test_collate_equals_reference_golden():
    """kitti_dataset.py:546-616: same tensors, same dtypes, and the host RNG stream left at the same position."""
    from umeregrobust_amd.datasets import batch_collate_fn_dset
    g = np.load(os.path.join(GOLD, "g10_collate.npz"))
    items = _items(g)
    for tag, data, nm, mx in (("b3", items, 60, 600), ("b1", items[1:2], 10000, 100000)):
        np.random.seed(10)
        res = batch_collate_fn_dset(data, num_matches=nm, max_pc_size=mx)
        assert len(res) == 11
        for name, v in zip(OUT, res):
            ref = g[f"{tag}_{name}"]
            assert v.numpy().dtype == ref.dtype and v.shape == ref.shape, name
            assert np.array_equal(v.numpy(), ref), name
        assert np.random.rand() == float(g[f"{tag}_next_rand"])
    # an injected generator gives the same draws as the global stream
    res2 = batch_collate_fn_dset(items, num_matches=60, max_pc_size=600, rng=np.random.RandomState(10))
    assert np.array_equal(res2[0].numpy(), g["b3_src_pts"]) and np.array_equal(res2[10].numpy(), g["b3_matches"])
    # matches survive the dilution consistently: they index the diluted clouds and point at the original partners
    m = res2[10][0].numpy()
    assert m[:, 0].max() < 600 and m[:, 1].max() < 600 and len(np.unique(m[:, 0])) == len(m)


def test_collate_carries_network_features_with_the_same_dilution():
    from umeregrobust_amd.datasets import batch_collate_fn_dset
    g = np.load(os.path.join(GOLD, "g10_collate.npz"))
    items = []
    for it in _items(g):
        sf = torch.arange(it[0].shape[0], dtype=torch.float32)[:, None].repeat(1, 4)     # feature = original index
        tf = torch.arange(it[3].shape[0], dtype=torch.float32)[:, None].repeat(1, 4)
        items.append(it + (sf, tf))
    res = batch_collate_fn_dset(items, num_matches=60, max_pc_size=600, rng=np.random.RandomState(10))
    assert len(res) == 13 and res[11].shape == (3, 600, 4) and res[12].shape == (3, 600, 4)
    for b in range(3):
        idx = res[11][b, :, 0].long()
        assert torch.equal(items[b][0][idx], res[0][b]) and torch.equal(items[b][3][res[12][b, :, 0].long()], res[4][b])
    assert np.array_equal(res[0].numpy(), g["b3_src_pts"])                                # the draws are the reference's


def test_cache_roundtrip_and_dataset(tmp_path):
    """kitti_dataset.py:441-458 / :647-657: file layout, keys, tuple order."""
    from umeregrobust_amd.datasets import CACHE_KEYS, CachedPairDataset, load_pickle, read_cached_pair, write_cached_pair
    g = np.load(os.path.join(GOLD, "g10_collate.npz"))
    items = _items(g)
    names = [(8, 0, 11), (8, 4, 15), (10, 7, 19)]
    for it, (s, a, b) in zip(items, names):
        write_cached_pair(os.path.join(tmp_path, "test", f"{s:02d}", f"{a:06d}_{b:06d}.pickle"), it)
    d = load_pickle(os.path.join(tmp_path, "test", "08", "000004_000015.pickle"))
    assert tuple(d.keys()) == CACHE_KEYS
    ds = CachedPairDataset(str(tmp_path), split="test")
    assert len(ds) == 3 and ds.files == names
    for i in range(3):
        got = ds[i]
        assert len(got) == 9 and all(torch.equal(x, y) for x, y in zip(got, items[i]))
    # nuScenes layout: the sequence directory name is the id (nuscenes_dataset.py:452)
    write_cached_pair(os.path.join(tmp_path, "ns", "test", "0103", "000002_000007.pickle"), items[0])
    dn = CachedPairDataset(os.path.join(tmp_path, "ns"), split="test", dataset="nuscenes")
    assert dn.files == [("0103", 2, 7)] and torch.equal(dn[0][0], items[0][0])
    with pytest.raises(KeyError, match="src_feat"):
        read_cached_pair(ds.path(0), with_features=True)
    write_cached_pair(ds.path(0), items[0], src_feat=torch.zeros(items[0][0].shape[0], 32), tgt_feat=torch.ones(items[0][3].shape[0], 32))
    got = read_cached_pair(ds.path(0), with_features=True)
    assert len(got) == 11 and got[9].shape == (items[0][0].shape[0], 32)
    bad = os.path.join(tmp_path, "bad.pickle")
    import pickle
    pickle.dump({"src_pts": 1}, open(bad, "wb"))
    with pytest.raises(KeyError, match="not a pair cache file"):
        read_cached_pair(bad)
    # the DataLoader + collate combination of evaluate.py:154-160
    from functools import partial
    from umeregrobust_amd.datasets import batch_collate_fn_dset
    np.random.seed(3)
    dl = torch.utils.data.DataLoader(ds, shuffle=False, num_workers=0, batch_size=1, drop_last=True,
                                     collate_fn=partial(batch_collate_fn_dset, num_matches=50, max_pc_size=500))
    batches = list(dl)
    assert len(batches) == 3 and batches[0][0].shape == (1, 500, 3) and batches[0][2].shape == (500, 4) and int(batches[0][2][:, 0].max()) == 0


def test_checkpoint_schema(tmp_path):
    """train_coloring.py:214-222 -> evaluate.py:164"""
    from umeregrobust_amd.datasets import checkpoint_state_dict
    sd = {"conv1.kernel": torch.zeros(27, 1, 32), "bn1.weight": torch.ones(32)}
    p = os.path.join(tmp_path, "w_checkpoint.pth")
    torch.save({"epoch": 3, "model_state_dict": sd, "optimizer_state_dict": {}, "total_loss": 0.1}, p)
    got = checkpoint_state_dict(p)
    assert set(got) == set(sd) and torch.equal(got["bn1.weight"], sd["bn1.weight"])
    assert set(checkpoint_state_dict(sd)) == set(sd)                       # save_model's bare state dict
    with pytest.raises(KeyError, match="optimizer_state_dict"):
        checkpoint_state_dict({"epoch": 1, "model_state_dict": sd, "total_loss": 0.0})
    with pytest.raises(TypeError):
        checkpoint_state_dict({"model_state_dict": {"a": 1}, "epoch": 0, "optimizer_state_dict": {}, "total_loss": 0})
