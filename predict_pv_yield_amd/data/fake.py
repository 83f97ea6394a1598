"""Synthetic whole-batch datasets: stand-in for nowcasting_dataloader.fake.FakeDataset + nowcasting_dataset's
Configuration (neither is installed).  Each dataset ITEM is a whole batch, as in the reference
(predict_pv_yield/data/dataloader.py:82-131: DataLoader(batch_size=None)).

Shapes follow the batch contract the path reads:
  satellite.data [B, C, T5, H, W] ~ N(0,1)         pv.pv_yield  [B, T5, 128] ~ U[0,1)
  gsp.gsp_yield  [B, T30, 32] ~ U[0,1)              nwp.data     [B, 10, T60, h, w] ~ N(0,1)
with T5 = history//5 + forecast//5 + 1, T30 = history//30 + forecast//30 + 1, T60 = ceil(history/60) + forecast//60 + 1.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from .batch import BatchML


@dataclass
class FakeDataConfiguration:
    batch_size: int = 32
    history_minutes: int = 60
    forecast_minutes: int = 30
    satellite_image_size_pixels: int = 64
    number_sat_channels: int = 11
    nwp_image_size_pixels: int = 2
    number_nwp_channels: int = 10
    n_pv_systems_per_example: int = 128
    n_gsp_per_example: int = 32
    seed: int = 1234
    sat_dtype: str = "float32"

    @property
    def seq_len_5(self):
        return self.history_minutes // 5 + self.forecast_minutes // 5 + 1

    @property
    def seq_len_30(self):
        return self.history_minutes // 30 + self.forecast_minutes // 30 + 1

    @property
    def seq_len_60(self):
        return int(np.ceil(self.history_minutes / 60)) + self.forecast_minutes // 60 + 1


def make_fake_batch(cfg: FakeDataConfiguration, generator: Optional[torch.Generator] = None, as_dict: bool = False):
    g = generator
    b = cfg.batch_size
    sat = torch.randn(b, cfg.number_sat_channels, cfg.seq_len_5, cfg.satellite_image_size_pixels,
                      cfg.satellite_image_size_pixels, generator=g)
    t0 = np.datetime64("2021-06-01T12:00", "ns").astype(np.int64)
    gsp_times = t0 + (np.arange(cfg.seq_len_30) - cfg.history_minutes // 30) * 30 * 60 * 10 ** 9
    sections = dict(
        metadata=dict(t0_datetime_utc=torch.full((b,), int(t0), dtype=torch.int64)),
        satellite=dict(data=sat),
        pv=dict(pv_yield=torch.rand(b, cfg.seq_len_5, cfg.n_pv_systems_per_example, generator=g),
                pv_system_row_number=torch.randint(0, 940, (b, cfg.n_pv_systems_per_example), generator=g)),
        gsp=dict(gsp_yield=torch.rand(b, cfg.seq_len_30, cfg.n_gsp_per_example, generator=g),
                 gsp_capacity=torch.rand(b, cfg.seq_len_30, cfg.n_gsp_per_example, generator=g) * 100 + 1,
                 gsp_id=torch.randint(1, 339, (b, cfg.n_gsp_per_example), generator=g),
                 gsp_datetime_index=torch.from_numpy(np.tile(gsp_times, (b, 1)))),
        nwp=dict(data=torch.randn(b, cfg.number_nwp_channels, cfg.seq_len_60, cfg.nwp_image_size_pixels,
                                  cfg.nwp_image_size_pixels, generator=g)),
    )
    return sections if as_dict else BatchML(**sections)


class FakeDataset(torch.utils.data.Dataset):
    """Each item is a whole (seeded, reproducible) batch; `length` batches per epoch."""

    def __init__(self, configuration: FakeDataConfiguration, length: int = 10, as_dict: bool = True):
        self.configuration = configuration
        self.length = length
        self.as_dict = as_dict

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        if idx >= self.length:
            raise IndexError(idx)
        g = torch.Generator().manual_seed(self.configuration.seed + idx)
        return make_fake_batch(self.configuration, g, as_dict=self.as_dict)
