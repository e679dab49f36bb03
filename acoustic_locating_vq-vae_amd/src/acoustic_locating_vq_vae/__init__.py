"""MI355X-native drop-in for the reference's ``acoustic_locating_vq_vae`` package (VQ-VAE hot path only)."""
