"""MI355X-native drop-in for the reference's `modules` package (hot path: util, keypoint_detector,
movement_embedding, dense_motion_module, generator; callers' helpers: discriminator, losses).

Sub-modules this package does not rebuild because they are outside the hot path -- `modules.prediction_module`, the GRU
of prediction.py:10 -- are found in the reference's own `modules/` directory when that tree is on sys.path behind this
package: the search path of the package is extended with every other `modules/` directory on sys.path, so
`from modules.prediction_module import PredictionModule` (and therefore `import run`) keeps working."""
import os as _os
import sys as _sys

_here = _os.path.abspath(_os.path.dirname(__file__))
for _p in list(_sys.path):
    _cand = _os.path.abspath(_os.path.join(_p or ".", "modules"))
    if _cand != _here and _os.path.isdir(_cand) and _cand not in __path__:
        __path__.append(_cand)
