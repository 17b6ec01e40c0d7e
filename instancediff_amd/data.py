"""Data side of the drivers, with the reference's batch layout.

* `SpeckleMedDataset` -- raw little-endian float32 LQ/GT images + `A_emb` files listed in a JSON
  (`dataset_file[phase] = [{A, B, A_emb, name}, ...]`), per-modality normalisation to [-1,1]
  (data/MedSpeckle.py:12-73; CT /1800 :55-61, cryo-EM /255 :63-67).  The reference hard-codes 224x224
  (:44-45); here the side length comes from `image_size` or is inferred from the file size.
* `SyntheticDataset` -- seeded synthetic pairs (utils/synthetic.py) for `mode: Synthetic` (the dataset itself is
  absent from the reference snapshot, .MISSING_LARGE_BLOBS).
* `DistIterSampler` -- epoch-seeded permutation, `indices[rank::world]` (data/data_sampler.py:46-61).
* `iterate_batches` -- minimal batcher producing {'LQ','GT','name','A_emb','LQ_path','GT_path'} dicts with pinned
  host tensors (n_workers: 0 in config.yml:37, so no worker processes are needed).
* `dump_raw` -- the LQ|pred|GT float32 `.raw` triptych of testUM.py:170-173.
"""
import json
import math
import os

import numpy as np
import torch

from .utils.synthetic import make_batch


class SpeckleMedDataset:
    def __init__(self, data_flist, phase="train", max_dataset_size=1000000, use_artifact_type=(), image_size=None):
        with open(data_flist, "r") as f:
            df = json.load(f)[phase]
        self.df = [it for it in df if it["name"] in use_artifact_type][:max_dataset_size]
        self.image_size = image_size

    def __len__(self):
        return len(self.df)

    def _read(self, path):
        a = np.fromfile(path, dtype=np.float32)
        s = self.image_size or int(round(math.sqrt(a.size)))
        return torch.from_numpy(a.reshape(1, s, s).copy())

    def __getitem__(self, index):
        it = self.df[index]
        A, B = self._read(it["A"]), self._read(it["B"])
        A_emb = torch.from_numpy(np.fromfile(it["A_emb"], dtype=np.float32).reshape(1, -1).copy())
        name = it["name"]
        if name == "scatter artifact in CT":
            A, B = A.clamp(0, 1800) / 1800.0, B.clamp(0, 1800) / 1800.0
        if name == "noise in cryo-EM image":
            A, B = A.clamp(0.0, 255.0) / 255.0, B.clamp(0.0, 255.0) / 255.0
        return {"LQ": A * 2.0 - 1.0, "GT": B * 2.0 - 1.0, "LQ_path": it["A"], "GT_path": it["B"], "name": name, "A_emb": A_emb}


class SyntheticDataset:
    def __init__(self, size=8, image_size=64, seed=1234):
        self.n, self.image_size, self.seed = size, image_size, seed
        b = make_batch(size, image_size, seed=seed, mixed=True)
        self.items = [{"LQ": b["input"][i], "GT": b["target"][i], "LQ_path": f"synthetic/{i}_lq.raw", "GT_path": f"synthetic/{i}_gt.raw",
                       "name": b["names"][i], "A_emb": b["A_emb"][i]} for i in range(size)]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.items[i]


def create_dataset(dataset_opt):
    mode = dataset_opt["mode"]
    if mode == "Synthetic":
        return SyntheticDataset(dataset_opt.get("max_dataset_size") or 8, dataset_opt.get("image_size") or 64, seed=1234)
    if mode == "SpeckleMed":
        return SpeckleMedDataset(dataset_opt["dataset_file"], phase=dataset_opt["phase"], max_dataset_size=dataset_opt["max_dataset_size"],
                                 use_artifact_type=dataset_opt["use_artifact_type"], image_size=dataset_opt.get("image_size"))
    raise NotImplementedError(f"Dataset [{mode}] is not recognized.")


class DistIterSampler:
    def __init__(self, dataset, num_replicas=1, rank=0, ratio=1):
        self.dataset, self.num_replicas, self.rank, self.epoch = dataset, num_replicas, rank, 0
        self.num_samples = int(math.ceil(len(dataset) * ratio / num_replicas))
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        g = torch.Generator().manual_seed(self.epoch)
        idx = [v % len(self.dataset) for v in torch.randperm(self.total_size, generator=g).tolist()]
        idx = idx[self.rank:self.total_size:self.num_replicas]
        assert len(idx) == self.num_samples
        return iter(idx)

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


def iterate_batches(dataset, batch_size, sampler=None, shuffle=False, seed=0, drop_last=False):
    if sampler is not None:
        order = list(iter(sampler))
    else:
        order = list(range(len(dataset)))
        if shuffle:
            order = torch.randperm(len(order), generator=torch.Generator().manual_seed(seed)).tolist()
    for s in range(0, len(order), batch_size):
        ids = order[s:s + batch_size]
        if drop_last and len(ids) < batch_size:
            break
        items = [dataset[i] for i in ids]
        out = {"LQ": torch.stack([it["LQ"] for it in items]), "GT": torch.stack([it["GT"] for it in items]),
               "A_emb": torch.stack([it["A_emb"] for it in items]), "name": [it["name"] for it in items],
               "LQ_path": [it["LQ_path"] for it in items], "GT_path": [it["GT_path"] for it in items]}
        if torch.cuda.is_available():
            for k in ("LQ", "GT", "A_emb"):
                out[k] = out[k].pin_memory()
        yield out


def dump_raw(path, lq, pred, gt):
    """float32 LQ | pred | GT side by side (testUM.py:170-173)"""
    arr = np.concatenate([np.asarray(lq, dtype=np.float32).squeeze(), np.asarray(pred, dtype=np.float32).squeeze(),
                          np.asarray(gt, dtype=np.float32).squeeze()], axis=-1)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    arr.tofile(path)
    return arr.shape
