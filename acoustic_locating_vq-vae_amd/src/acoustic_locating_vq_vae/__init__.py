"""MI355X-native drop-in for the hot path of the reference's ``acoustic_locating_vq_vae`` package.

This package overlays the VQ-VAE train-step modules (``vq_vae.*``), the dataset side that feeds them
(``data_preprocessing``, ``rir_dataset_generator.specsdataset`` + the new ``rir_dataset_generator.device_loader``;
SURVEY 8f rank 2) and the location head (``vq_vae.location_model``, rank 4).  Everything else the reference's
scripts import from the same package name -- ``visualization`` -- is outside that scope and is deliberately not
rebuilt here: ``extend_path`` lets such modules resolve from the reference checkout when it sits LATER on ``sys.path``
(PYTHONPATH=<this>:<this>/src:<reference>:<reference>/src), while every module that exists here wins.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
