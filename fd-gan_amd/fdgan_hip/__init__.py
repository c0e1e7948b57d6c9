"""fdgan_hip -- host side of the MI355X-native FD-GAN hot path.

`lib`     ctypes binding of libfdgan_hip.so (the C ABI in include/fdgan_hip.h)
`engine`  device buffers / NHWC views / op wrappers / plan recorder
"""
from . import lib, engine  # noqa: F401
