"""What ties a built kernel object to the sources it was built from (stdlib only: __graft_entry__.build() imports this
before the engine library exists).

`source_digest()` = the first 64 bits of SHA-256 over every kernel source of the engine (csrc/*.h, csrc/*.hip,
include/moog_engine.h) and the base hipcc flags.  It is compiled into libmoog_hip.so (`moog_source_digest()`) and into
every program-specialised step kernel (`moog_spec_source_digest()`, moog/_spec.py); the engine refuses a specialised
kernel whose digest is not its own (csrc/moog_engine.hip load_spec_kernel), so an object left behind by an older build of
the same ABI number is never picked up.  Objects also carry the digest as text (`MOOG_SRC_DIGEST=0x<16 hex digits>ull`), which
is how Python reads it without loading the object."""
import glob
import hashlib
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, '..', 'csrc'))
HEADER = os.path.normpath(os.path.join(_HERE, '..', '..', 'include', 'moog_engine.h'))
# the flags every translation unit of the engine is built with (the single place they are written down)
HIP_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-Wno-unused-value']
_MARK = re.compile(rb'MOOG_SRC_DIGEST=0x([0-9a-f]{16})ull')


def source_files():
    return sorted(glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.hip'))) + [HEADER]


def source_digest():
    h = hashlib.sha256()
    for p in source_files():
        h.update(os.path.basename(p).encode() + b'\0')
        with open(p, 'rb') as f:
            h.update(f.read())
        h.update(b'\0')
    h.update(' '.join(HIP_FLAGS).encode())
    return h.hexdigest()[:16]


def define_flag(digest=None):
    """The -D that compiles the digest into a translation unit."""
    return '-DMOOG_SRC_DIGEST=0x%sull' % (digest or source_digest())


def digest_of(path):
    """The digest a built object carries (its text marker), or None: no marker (an object of an older build) or no file."""
    try:
        with open(path, 'rb') as f:
            m = _MARK.search(f.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None
