"""spx_amd -- MI355X-native shifted proximal operators (the prox! hot path of
ShiftedProximalOperators.jl) behind the reference's own API.  The directory is named
`shiftedproximaloperators.jl_amd`; import it as `spx_amd` through the loader at the repository root."""
from . import _lib
from ._lib import SpxError
from .functions import (GroupNormL2, IndBallL0, NormL0, NormL1, NormL2, NormLinf, ProximableFunction,
                        RootNormLhalf)
from .sharding import shard_range
from .shifted import (ShiftedGroupNormL2, ShiftedGroupNormL2Binf, ShiftedIndBallL0, ShiftedIndBallL0BInf,
                      ShiftedNormL0, ShiftedNormL0Box, ShiftedNormL1, ShiftedNormL1B2, ShiftedNormL1Box,
                      ShiftedProximableFunction, ShiftedRootNormLhalf, ShiftedRootNormLhalfBox, context, device_values, iprox,
                      iprox_bang, prox,
                      prox_bang, prox_value, prox_value_bang, set_bounds_bang, set_radius_bang, shift_bang, shifted, synchronize, value)

__all__ = [n for n in dir() if not n.startswith("_")]
