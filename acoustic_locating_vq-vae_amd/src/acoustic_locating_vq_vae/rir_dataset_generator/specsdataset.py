"""``SpecsDataset`` -- a directory of ``{i}.pt`` samples plus ``dataset_config.npy``.

Reference: rir_dataset_generator/specsdataset.py:10-45 (written by scripts/genereate_dataset.py:97-103).  Same
constructor, attributes, ``__len__`` / ``__getitem__`` / ``get_source_coordinates``.  A sample file holds the 6-tuple
``(speech_spec, rir_spec, echoed_spec, sample_rate, theta, wiener_est)``: three (201, T) power spectrograms, an int,
a 1-element angle tensor and a (201,) Wiener estimate.
"""
import glob
import os

import numpy as np
import torch
from torch.utils.data import Dataset

CONFIG_KEYS = ("fs", "receiver_position", "room_dimensions", "reverberation_time", "n_sample", "R", "NFFT", "HOP_LENGTH",
               "Z_LOC_SOURCE")


class SpecsDataset(Dataset):
    def __init__(self, root_dir: str, transform=None):
        self.root_dir = root_dir
        self.transform = transform
        self.dataset_files = glob.glob(os.path.join(self.root_dir, "*.pt"))
        # the generator stores a plain dict with np.save, i.e. a pickled 0-d object array (genereate_dataset.py:103)
        cfg = np.load(os.path.join(root_dir, "dataset_config.npy"), allow_pickle=True).item()
        for key in CONFIG_KEYS:
            setattr(self, key, cfg[key])

    def __len__(self):
        return len(self.dataset_files)

    def item_path(self, idx):
        return os.path.join(self.root_dir, "{}.pt".format(idx))

    def __getitem__(self, idx):
        speech_spec, rir_spec, echoed_spec, sample_rate, theta, wiener_est = torch.load(self.item_path(idx))
        return speech_spec, rir_spec, echoed_spec, sample_rate, theta, wiener_est

    def get_source_coordinates(self, theta):
        """Source position(s) on the circle of radius R around the receiver, clipped to the room (:38-45)."""
        ring = np.stack((self.R * np.cos(theta).T, self.R * np.sin(theta).T, np.array([self.Z_LOC_SOURCE])), axis=1)
        return np.minimum(self.receiver_position + ring, self.room_dimensions)
