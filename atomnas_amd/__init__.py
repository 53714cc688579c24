"""atomnas_amd: MI355X-native (gfx950) implementation of the AtomNAS supernet-training hot path.

Host code mirrors the reference's Python interface (models.mobilenet_supernet / mobilenet_base, utils.prune / optim /
rmsprop / distributed, train.py + yaml configs); the arithmetic runs in hand-written HIP kernels behind the C ABI declared
in include/atomnas_hip.h.  There is no CPU fallback in this package.
"""
__version__ = "0.1.0"
