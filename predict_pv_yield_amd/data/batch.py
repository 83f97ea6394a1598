"""BatchML-shaped batch container (nowcasting_dataloader.batch.BatchML is not installed here).

Only the fields the hot path reads are modelled:
  satellite.data [B,C,T,H,W]; pv.pv_yield [B,T5,n_pv]; pv.pv_system_row_number [B,n_pv];
  gsp.gsp_yield [B,T30,n_gsp]; gsp.gsp_datetime_index; gsp.gsp_capacity; gsp.gsp_id;
  nwp.data [B,C,T60,h,w]; metadata.t0_datetime_utc.
`BatchML(**dict)` and `batch["pv_yield"]`-style subscripting (model.py:131,141) both work.
"""
from types import SimpleNamespace
from typing import Any, Dict

import torch


class _Section(SimpleNamespace):
    def to(self, device):
        return _Section(**{k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in vars(self).items()})

    def as_dict(self) -> Dict[str, Any]:
        return dict(vars(self))


_ALIASES = {"pv_yield": ("pv", "pv_yield"), "gsp_yield": ("gsp", "gsp_yield"), "nwp": ("nwp", "data"),
            "satellite": ("satellite", "data")}


class BatchML:
    SECTIONS = ("metadata", "satellite", "pv", "gsp", "nwp", "sun", "datetime", "topographic")

    def __init__(self, **sections):
        for name in self.SECTIONS:
            sec = sections.get(name)
            if isinstance(sec, dict):
                sec = _Section(**sec)
            setattr(self, name, sec)
        extra = set(sections) - set(self.SECTIONS)
        if extra:
            raise TypeError(f"BatchML: unknown sections {sorted(extra)}")

    def __getitem__(self, key: str):
        if key in _ALIASES:
            sec, field = _ALIASES[key]
            return getattr(getattr(self, sec), field)
        return getattr(self, key)

    def to(self, device) -> "BatchML":
        out = BatchML()
        for name in self.SECTIONS:
            sec = getattr(self, name)
            setattr(out, name, sec.to(device) if sec is not None else None)
        return out

    def as_dict(self) -> Dict[str, Dict[str, Any]]:
        return {n: getattr(self, n).as_dict() for n in self.SECTIONS if getattr(self, n) is not None}
