"""MI355X-native captioning decoder hot path (prefix join -> TransformerDecoder
forward/backward -> fused vocabulary projection + cross-entropy -> greedy decoding).

The arithmetic lives in libcapdec.so (hand-written HIP for gfx950, C ABI in
include/capdec.h); this package is the host-side mirror of the reference's
Python interface.  There is no CPU fallback.
"""
__version__ = "0.1.0"
