"""tsar-mvs_amd — MI355X-native PatchMatch-MVS matcher (host-side Python mirror of the C ABI in
include/tsar.h).  The compute lives in csrc/ (hand-written HIP for gfx950, built into
libtsar_hip.so); this package only loads it and marshals buffers."""
__version__ = "0.1.0"
