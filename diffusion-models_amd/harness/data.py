"""The training data path with the data set resident in HBM (reference: train.py:109-138 load_dataset,
train.py:312-329 get_train_dataloader, utils/augmentation.py).

The reference decodes every image on the host in DataLoader worker processes (PIL -> ToTensor -> Normalize, optional
random flips) and copies each batch to the device. CIFAR-10 / MNIST are 150 / 47 MB as uint8: here the whole set
lives in device memory as the uint8 array torchvision exposes (`dataset.data`, [N, H, W, C] or [N, H, W]) and a batch
is ONE kernel: gather by index, mirror, normalise to fp32 NCHW - the same numbers the reference's transform produces,
bit for bit (each fp32 operation rounded separately).

Order of the batches = the reference's: DataLoader(shuffle=True, generator=manual_seed(seed)) draws, per epoch, one
int64 "base seed" and then torch.randperm(N) from that generator (plus one discarded permutation at the epoch's end). The flip decisions of the reference come from
per-worker CPU generators; here they come from a torch generator on the device (same distribution: each listed flip
independently with p = 0.5).
"""
from typing import Iterator, List, Optional, Tuple

import torch

from dm_hip import check, lib

_FLIP_BITS = {"RandomHorizontalFlip": 1, "RandomVerticalFlip": 2}   # utils/augmentation.py:6-9


def epoch_permutation(n: int, generator: torch.Generator) -> torch.Tensor:
    """The index order of one epoch of DataLoader(shuffle=True, generator=generator): the loader's iterator first
    draws its base seed from the generator, then RandomSampler draws torch.randperm(n) - and, when the epoch's indices
    are used up, one more permutation for its (empty) remainder, which advances the generator all the same."""
    torch.empty((), dtype=torch.int64).random_(generator=generator)
    perm = torch.randperm(n, generator=generator)
    torch.randperm(n, generator=generator)
    return perm


class DeviceDataset:
    """uint8 images + int64 labels in device memory; batches come out normalised, fp32, NCHW."""

    def __init__(self, data_u8: torch.Tensor, labels: torch.Tensor, device, augmentations: Optional[List[str]] = None,
                 normalize: str = "transform", seed: int = 42):
        if data_u8.dtype != torch.uint8 or data_u8.dim() not in (3, 4):
            raise ValueError("DeviceDataset: data must be uint8 [N, H, W, C] or [N, H, W]")
        if data_u8.dim() == 3:
            data_u8 = data_u8[..., None]
        for a in augmentations or []:
            assert a in _FLIP_BITS, f"unknown aug type {a}"   # utils/augmentation.py:15
        if normalize not in ("transform", "data_tensor"):
            raise ValueError(f"unknown normalisation {normalize}")
        self.device = torch.device(device)
        self.data = data_u8.contiguous().to(self.device)
        self.labels = labels.to(torch.int64).to(self.device)
        self.N, self.H, self.W, self.C = self.data.shape
        self.flips = [_FLIP_BITS[a] for a in (augmentations or [])]
        self.mode = 0 if normalize == "transform" else 1
        self.flip_gen = torch.Generator(device=self.device).manual_seed(seed)

    def __len__(self):
        return self.N

    def batch(self, idx: torch.Tensor, flip: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(x fp32 [B, C, H, W], labels int64 [B]) for the rows idx. flip: uint8 [B] (bit 0 horizontal, bit 1
        vertical); None draws the configured random flips."""
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= self.N):
            raise IndexError("DeviceDataset.batch: index out of range")
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        B = idx.numel()
        if flip is None and self.flips:
            flip = torch.zeros(B, dtype=torch.uint8, device=self.device)
            for bit in self.flips:   # each flip independently with p = 0.5 (torchvision's default)
                flip |= (torch.rand(B, device=self.device, generator=self.flip_gen) < 0.5).to(torch.uint8) * bit
        elif flip is not None:
            flip = flip.to(device=self.device, dtype=torch.uint8).contiguous()
        out = torch.empty((B, self.C, self.H, self.W), dtype=torch.float32, device=self.device)
        check(lib.dm_u8_batch(self.data.data_ptr(), idx.data_ptr(), flip.data_ptr() if flip is not None else None,
                              out.data_ptr(), B, self.C, self.H, self.W, self.N, self.mode,
                              torch.cuda.current_stream().cuda_stream))
        return out, self.labels[idx]

    def data_tensor(self, chunk: int = 8192) -> torch.Tensor:
        """The whole set as fp32 [N, C, H, W] scaled by v / 127.5 - 1 (train.py:124-125): the reference rows of the
        memorisation metric."""
        mode, self.mode = self.mode, 1
        try:
            parts = [self.batch(torch.arange(i, min(self.N, i + chunk)), flip=torch.zeros(min(chunk, self.N - i), dtype=torch.uint8))[0]
                     for i in range(0, self.N, chunk)]
        finally:
            self.mode = mode
        return torch.cat(parts, dim=0)

    def epochs(self, batch_size: int, seed: int, drop_last: bool = False, start_epoch: int = 0,
               start_step: int = 0) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        """Endless stream of batches in the reference DataLoader's order (train.py:316-327).
        start_epoch / start_step: take the stream up after that many whole epochs and that many batches of the next
        one (a resumed run: the `epochs` / `steps_in_epoch` of latest.pt). The reference replays its DataLoader to get
        there (skip_previous_steps, train.py:598-605); here the skipped epochs only advance the shuffle generator and
        the skipped batches are never gathered. The flip generator is NOT advanced: a resumed run restores its state
        from latest.pt (checkpoint.load_latest_ckpt(..., dataset=...))."""
        rng = torch.Generator().manual_seed(seed)
        for _ in range(start_epoch):
            epoch_permutation(self.N, rng)
        first = start_step * batch_size
        while True:
            perm = epoch_permutation(self.N, rng)
            for i in range(first, self.N, batch_size):
                idx = perm[i:i + batch_size]
                if drop_last and idx.numel() < batch_size:
                    break
                yield self.batch(idx)
            first = 0
