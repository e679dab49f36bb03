"""CPU restatement of the reference's dataset collate -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/src/acoustic_locating_vq_vae/data_preprocessing.py line by line (the module itself cannot be
imported here: its top-level ``import rir_generator`` names a C++ package that is absent).  The reference
holds no fixtures for these functions: PARITY UNPINNED beyond this reading of the source.
"""
import numpy as np
import torch


def combine_tensors_with_min_dim(tensor_list):                      # data_preprocessing.py:18-52
    if not tensor_list:
        raise ValueError("Input tensor list cannot be empty")
    H = tensor_list[0].shape[1]
    for tensor in tensor_list:
        if tensor.shape[1] != H:
            raise ValueError("All tensors in the list must have the same height (H)")
    min_dim = min(tensor.shape[2] for tensor in tensor_list)
    combined = torch.zeros((len(tensor_list), H, min_dim), dtype=torch.complex64)
    for i, tensor in enumerate(tensor_list):
        combined[i, :, :] = tensor[:, :, :min_dim]
    return combined


def spec_dataset_preprocessing(data):                               # data_preprocessing.py:55-89
    cols = [[] for _ in range(6)]
    for (speech_spec, rir_spec, echoed_spec, sample_rate, theta, wiener_est) in data:
        if speech_spec.shape[1] < 500:
            continue
        cols[0].append(speech_spec[:, :500])
        cols[1].append(rir_spec[:, :500])
        cols[2].append(echoed_spec[:, :500])
        cols[3].append(torch.as_tensor(sample_rate))
        cols[4].append(theta)
        cols[5].append(wiener_est)
    if len(cols[0]) == 0:
        return [], [], [], [], [], []
    return tuple(torch.stack(c) for c in cols)


def source_coordinates(theta, R, z_loc_source, receiver_position, room_dimensions):   # specsdataset.py:38-45
    z_loc = np.array([z_loc_source])
    h = receiver_position + np.stack((R * np.cos(theta).T, R * np.sin(theta).T, z_loc), axis=1)
    return np.minimum(h, room_dimensions)


def make_synthetic_dataset(root, n, lengths, seed=0, bins=201):
    """Write ``n`` samples in the generator's file format (genereate_dataset.py:97-103) with closed-form content:
    sample i has ``lengths[i % len(lengths)]`` frames.  Returns the list of 6-tuples written."""
    import os
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(seed)
    items = []
    for i in range(n):
        T = lengths[i % len(lengths)]
        spec = lambda: torch.from_numpy(rng.random((bins, T), dtype=np.float32) ** 2)   # noqa: E731
        item = (spec(), spec(), spec(), 16000, torch.from_numpy(rng.uniform(-np.pi, np.pi, size=1)),
                torch.from_numpy(rng.random(bins, dtype=np.float32)))
        torch.save(item, os.path.join(root, "%d.pt" % i))
        items.append(item)
    cfg = {"fs": 16000, "receiver_position": [2.5, 1.5, 1.5], "room_dimensions": [4, 5, 3], "reverberation_time": 0.4,
           "n_sample": 6400, "R": 1, "NFFT": 400, "HOP_LENGTH": 160, "Z_LOC_SOURCE": 1}
    np.save(os.path.join(root, "dataset_config.npy"), cfg)
    return items
