"""DeviceLoader on the GPU: resident and streaming batches equal the CPU collate of the same samples, and a train step
consumes them directly."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import data_oracle as DO  # noqa: E402


@pytest.mark.parametrize("resident", [True, False])
def test_device_loader_feeds_the_trainer(tmp_path, resident):
    from acoustic_locating_vq_vae.rir_dataset_generator.device_loader import DeviceLoader
    from acoustic_locating_vq_vae.rir_dataset_generator.specsdataset import SpecsDataset
    from acoustic_locating_vq_vae.train_step import Trainer
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    root = str(tmp_path / "specs")
    items = DO.make_synthetic_dataset(root, 12, [501, 480, 620, 500], seed=4)
    loader = DeviceLoader(SpecsDataset(root), 4, shuffle=True, device="cuda", resident=resident, workers=4, seed=2)
    torch.manual_seed(3)
    model = ConvolutionalVQVAE(201, 32, 8, 2, 16, 0.25, 32).cuda().train()
    trainer = Trainer(model, "speech")
    for _ in range(4):
        batch = next(iter(loader))
        want = DO.spec_dataset_preprocessing([items[i] for i in loader.last_indices])
        for got, ref in zip(batch, want):
            assert got.is_cuda and torch.equal(got.cpu(), ref)
        loss, recon_error, perplexity = trainer.step(batch[0])          # speech_specs, as train_speech.py:59-62
        assert torch.isfinite(loss)
    loader.close()
