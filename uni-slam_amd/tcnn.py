"""
`import unislam_amd.tcnn as tcnn` -- the two tinycudann entry points Uni-SLAM uses (src/UNISLAM.py:242,
src/networks/decoders.py:50,61), served by the gfx950 kernels.  See INTEGRATION.md.
"""
from .hashgrid import HashGridEncoding as Encoding  # noqa: F401
from .network import FusedMLP as Network  # noqa: F401
