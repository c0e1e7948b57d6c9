"""Identity of the sources a libfdgan_hip.so was built from (stdlib only: __graft_entry__.build() loads this file by path).

The library is git-ignored and rebuilt by mtime, and the hand-bumped FDGAN_ABI_VERSION does not notice a changed argument
struct: a stale library would load silently and run kernels against the wrong layout.  build() therefore embeds the hash of
every file the library is compiled from (`fdgan_build_id()`), and lib.load() compares it with the hash of the same files
whenever they lie next to the package (always true in the repository and on the GPU box's snapshot) and refuses to load on
a mismatch.
"""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
HEADER = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "fdgan_hip.h")


def source_files(csrc=CSRC, header=HEADER):
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    return files + [header]


def source_build_id(csrc=CSRC, header=HEADER):
    """16 hex digits over (file name, contents) of csrc/*.hip, csrc/*.h and include/fdgan_hip.h; None when the sources are
    not there (an installed library without its tree: nothing to compare with)."""
    if not os.path.isdir(csrc) or not os.path.exists(header):
        return None
    h = hashlib.sha256()
    for path in source_files(csrc, header):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]
