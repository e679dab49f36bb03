"""the ``{i}.pt`` dataset format, its collate and the persistent loader -- host logic, CPU only.
Checked against the line-by-line restatement in oracle/data_oracle.py on synthetic files in the generator's format."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import data_oracle as DO  # noqa: E402

LENGTHS = [501, 480, 620, 500, 499, 733]


@pytest.fixture(scope="module")
def dataset_dir(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("specs"))
    items = DO.make_synthetic_dataset(root, 14, LENGTHS, seed=3)
    return root, items


def same_batch(a, b):
    assert len(a) == len(b) == 6
    for x, y in zip(a, b):
        assert torch.equal(torch.as_tensor(x), torch.as_tensor(y)) and x.dtype == y.dtype and x.shape == y.shape


def test_specs_dataset_fields_items_and_source_coordinates(dataset_dir):
    from acoustic_locating_vq_vae.rir_dataset_generator.specsdataset import SpecsDataset
    root, items = dataset_dir
    ds = SpecsDataset(root)
    assert len(ds) == 14 and ds.transform is None and ds.root_dir == root
    assert (ds.fs, ds.NFFT, ds.HOP_LENGTH, ds.R, ds.Z_LOC_SOURCE, ds.n_sample) == (16000, 400, 160, 1, 1, 6400)
    assert ds.receiver_position == [2.5, 1.5, 1.5] and ds.room_dimensions == [4, 5, 3] and ds.reverberation_time == 0.4
    got = ds[5]
    for g, w in zip(got, items[5]):
        assert (torch.equal(g, w) if torch.is_tensor(w) else g == w)
    theta = np.array([2.2])
    want = DO.source_coordinates(theta, ds.R, ds.Z_LOC_SOURCE, ds.receiver_position, ds.room_dimensions)
    assert np.array_equal(ds.get_source_coordinates(theta), want) and want.shape == (1, 3)


def test_collate_matches_reference_rules(dataset_dir):
    from acoustic_locating_vq_vae import data_preprocessing as P
    _, items = dataset_dir
    same_batch(P.spec_dataset_preprocessing(items), DO.spec_dataset_preprocessing(items))     # drops 480 / 499, crops
    speech = P.spec_dataset_preprocessing(items)[0]
    assert speech.shape == (9, 201, 500) and torch.equal(speech[0], items[0][0][:, :500])
    short = [it for it in items if it[0].shape[1] < 500]
    out = P.spec_dataset_preprocessing(short)                                                  # nothing qualifies
    assert all(o == [] for o in out) and len(out) == 6
    stack = [torch.randn(1, 7, n) for n in (9, 5, 12)]
    got, want = P.combine_tensors_with_min_dim(stack), DO.combine_tensors_with_min_dim(stack)
    assert got.dtype == torch.complex64 and torch.equal(got, want) and got.shape == (3, 7, 5)
    with pytest.raises(ValueError, match="cannot be empty"):
        P.combine_tensors_with_min_dim([])
    with pytest.raises(ValueError, match="same height"):
        P.combine_tensors_with_min_dim([torch.randn(1, 7, 4), torch.randn(1, 6, 4)])
    libri = [(torch.randn(7, n), 0, 0, 0, 0, 16000 + n) for n in (9, 5)]
    specs, sr = P.batchify_spectrograms(libri, 400, 160)
    assert specs.shape == (2, 7, 5) and sr == 16005


@pytest.mark.parametrize("resident", [True, False])
def test_device_loader_is_a_persistent_stream_of_collated_batches(dataset_dir, resident):
    from acoustic_locating_vq_vae.rir_dataset_generator.device_loader import DeviceLoader
    from acoustic_locating_vq_vae.rir_dataset_generator.specsdataset import SpecsDataset
    root, items = dataset_dir
    ds = SpecsDataset(root)
    loader = DeviceLoader(ds, 4, shuffle=False, device="cpu", resident=resident, workers=0 if not resident else 3)
    assert iter(loader) is loader and loader.resident == resident
    seen = []
    for _ in range(5):
        batch = next(iter(loader))                     # the call pattern of the reference's loops
        ids = loader.last_indices
        if resident:                                   # short samples were dropped at load time: full batches
            assert len(ids) == 4 and all(items[i][0].shape[1] >= 500 for i in ids)
        same_batch(batch, DO.spec_dataset_preprocessing([items[i] for i in ids]))
        seen += ids
    assert seen[:8] == ([0, 2, 3, 5, 6, 8, 9, 11] if resident else list(range(8)))   # sequential order, wraps around
    loader.close()
    shuffled = DeviceLoader(ds, 3, shuffle=True, device="cpu", resident=True, seed=1)
    epoch = []
    for _ in range(3):                                 # 9 samples of >= 500 frames = one epoch of 3 batches
        next(shuffled)
        epoch += shuffled.last_indices
    assert sorted(epoch) == sorted(shuffled.sample_ids) and epoch != sorted(epoch)    # a permutation per epoch
    shuffled.close()
