"""Dataset side of the train loops: the ``{i}.pt`` sample format, its collate and the loaders.

Overlays ``specsdataset`` and adds ``device_loader``; any other module of the reference's sub-package keeps resolving
from the reference checkout later on the path (same ``extend_path`` arrangement as the parent package).
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
