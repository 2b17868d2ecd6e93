"""
Compatibility module: the autograd-free drivers of the hot loops live in mapstep.py (MapStep: src/Mapper.py:366-445),
window.py (MapWindow: the same with the window's camera poses on the device, joint_opt) and trackstep.py (TrackStep:
src/Tracker.py:149-244).
"""
from .mapstep import MapStep, _align          # noqa: F401
from .trackstep import TrackStep              # noqa: F401
