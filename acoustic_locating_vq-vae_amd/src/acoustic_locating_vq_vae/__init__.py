"""MI355X-native drop-in for the hot path of the reference's ``acoustic_locating_vq_vae`` package.

This package overlays ONLY the VQ-VAE train-step modules (``vq_vae.*``).  Everything else the reference's scripts
import from the same package name -- ``data_preprocessing``, ``rir_dataset_generator.specsdataset``,
``visualization``, ``vq_vae.location_model`` -- is host-side code outside the hot path and is deliberately not
rebuilt here: ``extend_path`` lets those modules resolve from the reference checkout when it sits LATER on
``sys.path`` (PYTHONPATH=<this>:<this>/src:<reference>:<reference>/src), while every module that exists here wins.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
