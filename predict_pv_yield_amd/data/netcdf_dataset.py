"""Whole-batch dataset over prepared batch files + device-side input step — mirror of the (moved) reference loader
predict_pv_yield/netcdf_dataset.py:12-119 and of the way predict_pv_yield/data/dataloader.py:82-131 consumes it.

  * one FILE = one whole BATCH (`__getitem__(batch_idx)` returns it; the DataLoader uses batch_size=None);
  * satellite imagery is stored as raw int16 counts and normalised `(x.astype(f32) - SAT_MEAN) / SAT_STD` per channel
    (netcdf_dataset.py:96-101).  Here the int16 tensor is what crosses PCIe (half the bytes of f32) and the normalisation
    runs on the MI355X (pv_normalise_i16), bit-identical to the host formula;
  * `DeviceBatchPrefetcher` keeps `depth` batches in flight: a reader thread loads + pins the next file while the GPU
    trains, the host->device copy and the normalisation run on a side stream, the consumer only waits on an event.

File formats: `.npz` with flat keys "<section>/<field>" (always available), or classic NetCDF-3 `.nc` with variables
"<section>__<field>" through scipy.io (NetCDF-4/HDF5, xarray, gcsfs/s3fs are not installed in this image: cloud download
and NetCDF-4 decoding stay outside the hot path).
"""
import os
import queue
import threading
from typing import Dict, Optional

import numpy as np
import torch

from .batch import BatchML

SAT_VARIABLE_NAMES = ("HRV", "IR_016", "IR_039", "IR_087", "IR_097", "IR_108", "IR_120", "IR_134", "VIS006", "VIS008",
                      "WV_062", "WV_073")
# netcdf_dataset.py:19-32
SAT_MEAN = np.array([93.23458, 131.71373, 843.7779, 736.6148, 771.1189, 589.66034, 862.29816, 927.69586, 90.70885,
                     107.58985, 618.4583, 532.47394], dtype=np.float32)
SAT_STD = np.array([115.34247, 139.92636, 36.99538, 57.366386, 30.346825, 149.68007, 51.70631, 35.872967, 115.77212,
                    120.997154, 98.57828, 99.76469], dtype=np.float32)


def get_netcdf_filename(batch_idx: int, ext: str = "npz") -> str:
    """nowcasting_dataset.utils.get_netcdf_filename: zero-padded batch index."""
    return f"{batch_idx:06d}.{ext}"


def write_batch_file(path: str, sections: Dict[str, Dict[str, np.ndarray]]) -> None:
    """Writes one whole batch ({section: {field: array}}, the BatchML layout) as .npz or NetCDF-3 (.nc)."""
    flat = {f"{s}/{k}": np.asarray(v) for s, d in sections.items() for k, v in d.items()}
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    if path.endswith(".npz"):
        np.savez(path, **flat)
        return
    from scipy.io import netcdf_file
    with netcdf_file(path, "w") as f:
        for key, arr in flat.items():
            if arr.dtype == np.int64:
                arr = arr.astype(np.float64)          # classic NetCDF has no 64-bit integers
            name = key.replace("/", "__")
            dims = []
            for ax, n in enumerate(arr.shape):
                dim = f"{name}_d{ax}"
                f.createDimension(dim, n)
                dims.append(dim)
            var = f.createVariable(name, arr.dtype.newbyteorder("=").char if arr.dtype.kind != "i" else arr.dtype.str[1:],
                                   tuple(dims))
            var[:] = arr


def read_batch_file(path: str) -> Dict[str, Dict[str, torch.Tensor]]:
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    if path.endswith(".npz"):
        with np.load(path) as z:
            items = {k: z[k] for k in z.files}
        sep = "/"
    else:
        from scipy.io import netcdf_file
        with netcdf_file(path, "r", mmap=False) as f:
            items = {k: np.array(v[:]) for k, v in f.variables.items()}
        sep = "__"
    for key, arr in items.items():
        section, field = key.split(sep, 1)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("="))
        out.setdefault(section, {})[field] = torch.from_numpy(np.ascontiguousarray(arr))
    return out


class NetCDFDataset(torch.utils.data.Dataset):
    """Loads the batches saved by the data-preparation step; every item is a whole batch (dict of sections)."""

    def __init__(self, n_batches: int, src_path: str, tmp_path: Optional[str] = None, configuration=None):
        self.n_batches, self.src_path, self.tmp_path, self.configuration = n_batches, src_path, tmp_path, configuration

    def per_worker_init(self, worker_id: int):
        pass      # the reference opens its cloud filesystem here; local files need nothing

    def __len__(self):
        return self.n_batches

    def __getitem__(self, batch_idx: int):
        if not 0 <= batch_idx < self.n_batches:
            raise IndexError(f"batch_idx must be in the range [0, {self.n_batches}), not {batch_idx}!")
        for ext in ("npz", "nc"):
            path = os.path.join(self.src_path, get_netcdf_filename(batch_idx, ext))
            if os.path.exists(path):
                return read_batch_file(path)
        raise FileNotFoundError(os.path.join(self.src_path, get_netcdf_filename(batch_idx, "npz|nc")))


def worker_init_fn(worker_id):
    info = torch.utils.data.get_worker_info()
    if info is not None:
        info.dataset.per_worker_init(info.id)


def satellite_channel_stats(n_channels: int, device):
    """SAT_MEAN / SAT_STD for the channels of a batch: all 12, or the 11 non-HRV ones (SAT_VARIABLE_NAMES[1:])."""
    sl = slice(0, 12) if n_channels == 12 else slice(1, 1 + n_channels)
    return torch.from_numpy(SAT_MEAN[sl]).to(device), torch.from_numpy(SAT_STD[sl]).to(device)


def normalise_satellite_on_device(batch):
    """If batch.satellite.data holds raw int16 counts [B, C, T, H, W] on the MI355X, replace it by the normalised f32
    tensor (netcdf_dataset.py:96-101) computed by pv_normalise_i16.  Works on BatchML or on the dict form."""
    sat = batch["satellite"]["data"] if isinstance(batch, dict) else batch.satellite.data
    if sat.dtype != torch.int16:
        return batch
    if not sat.is_cuda:
        raise RuntimeError("normalise_satellite_on_device: move the batch to the MI355X first (no CPU path)")
    from .. import hip_ops as K
    b, c, t, h, w = sat.shape
    mean, std = satellite_channel_stats(c, sat.device)
    out = K.normalise(sat.contiguous(), mean, std, inner=t * h * w)
    if isinstance(batch, dict):
        batch["satellite"]["data"] = out
    else:
        batch.satellite.data = out
    return batch


class DeviceBatchPrefetcher:
    """Iterates a loader of HOST whole-batches and yields DEVICE batches, `depth` ahead of the consumer: a reader thread
    fetches and pins the next batch; its host->device copies and the int16 normalisation are queued on a side stream and
    guarded by an event the consumer's stream waits on (the training stream never blocks on PCIe)."""

    def __init__(self, loader, device, depth: int = 2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, depth)

    def __len__(self):
        return len(self.loader)

    def _stage(self, host_batch, stream):
        batch = host_batch if isinstance(host_batch, BatchML) else BatchML(**host_batch)
        with torch.cuda.stream(stream):
            pinned = BatchML()
            for name in BatchML.SECTIONS:
                sec = getattr(batch, name)
                if sec is None:
                    continue
                moved = {k: (v.pin_memory().to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v)
                         for k, v in vars(sec).items()}
                setattr(pinned, name, type(sec)(**moved))
            normalise_satellite_on_device(pinned)
            done = stream.record_event()
        return pinned, done

    def __iter__(self):
        if self.device.type != "cuda":
            raise RuntimeError("DeviceBatchPrefetcher feeds the MI355X (no CPU path)")
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stream = torch.cuda.Stream(device=self.device)
        stop = threading.Event()

        def reader():
            try:
                torch.cuda.set_device(self.device)
                for host_batch in self.loader:
                    if stop.is_set():
                        break
                    q.put(self._stage(host_batch, stream))
                q.put(None)
            except BaseException as e:      # surfaced in the consumer
                q.put(e)

        th = threading.Thread(target=reader, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                batch, done = item
                consumer = torch.cuda.current_stream(self.device)
                consumer.wait_event(done)
                # the tensors were allocated under the side stream: tell the caching allocator that the consumer's stream
                # uses them too, otherwise a block freed by the consumer is handed back to the side-stream pool at once
                # and the reader thread may overwrite it (batch N+depth+1's copy) while kernels of batch N still read it
                for name in BatchML.SECTIONS:
                    sec = getattr(batch, name)
                    if sec is None:
                        continue
                    for v in vars(sec).values():
                        if isinstance(v, torch.Tensor) and v.is_cuda:
                            v.record_stream(consumer)
                yield batch
        finally:
            stop.set()
            while not q.empty():
                q.get_nowait()
