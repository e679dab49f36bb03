from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)   # vq_vae.location_model keeps resolving from the reference checkout
