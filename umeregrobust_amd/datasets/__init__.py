"""Input side of the path: the reference's cached-pair file format and loader output contract."""
from .kitti_dataset import (CACHE_KEYS, CachedPairDataset, batch_collate_fn_dset, checkpoint_state_dict, load_pickle,  # noqa: F401
                            read_cached_pair, sparse_collate, write_cached_pair)
