"""DataModule for the dict batches of experiments/003_perceiver_processes_single_sat_image_then_rnn.py (synthetic: the
experiment's own loaders read the OCF zarr stores, outside the hot path).  Whole batches per item, equal shards
per rank (padded by wrapping) like data/dataloader.py."""
import torch

from ..distributed import shard_indices
from ..lightning import LightningDataModule
from ..models.perceiver.exp003 import FakeExp003Dataset


class _Slice(torch.utils.data.Dataset):
    def __init__(self, base, indices):
        self.base, self.indices = base, list(indices)

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        if i >= len(self):
            raise IndexError(i)
        return self.base[self.indices[i]]


class Exp003DataModule(LightningDataModule):
    def __init__(self, batch_size: int = 32, image_size_pixels: int = 128, n_train_data: int = 8, n_val_data: int = 2,
                 seed: int = 1234):
        super().__init__()
        self.batch_size, self.image_size_pixels = batch_size, image_size_pixels
        self.n_train_data, self.n_val_data, self.seed = n_train_data, n_val_data, seed

    def _loader(self, n, seed):
        ds = _Slice(FakeExp003Dataset(self.batch_size, self.image_size_pixels, length=n, seed=seed), shard_indices(n))
        return torch.utils.data.DataLoader(ds, batch_size=None, num_workers=0)

    def train_dataloader(self):
        return self._loader(self.n_train_data, self.seed)

    def val_dataloader(self):
        return self._loader(self.n_val_data, self.seed + 100000)

    def test_dataloader(self):
        return self._loader(self.n_val_data, self.seed + 200000)
