"""skfusion_amd -- MI355X-native engine for the DFMF / DFMC collective tri-factorisation loop
behind scikit-fusion's own API (``fusion.Dfmf``, ``fusion.Dfmc``, ``fusion.DfmfTransform``,
``FusionGraph`` / ``Relation`` / ``ObjectType``).

The arithmetic runs in ``lib/libskfusion_hip.so`` (hand-written gfx950 kernels behind the C
ABI of ``include/skfusion_hip.h``); this package is the host-side mirror of the reference's
Python interface for that path.  There is no CPU fallback: without the HIP library and a GPU
``fuse()`` / ``transform()`` raise.
"""
from . import fusion                                     # noqa: F401
from .fusion import *                                    # noqa: F401,F403

__version__ = '0.1.0'
