"""MI355X-native drop-in for the per-slice destripe hot path of aind-smartspim-destripe."""

__version__ = "0.1.0"
