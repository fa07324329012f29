"""MI355X-native reverse-diffusion sampler for compression-artifact restoration."""
__version__ = "0.1.0"
