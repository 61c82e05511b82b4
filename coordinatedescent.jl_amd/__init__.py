"""coordinatedescent.jl_amd -- MI355X-native coordinate-descent sweep behind
CoordinateDescent.jl's operator interface.

The directory name contains a dot, so import it through the root-level shim:
``import coordinatedescent_jl_amd as cd``.

  csrc/        hand-written HIP kernels for gfx950 + the C-ABI (include/cdhip.h)
  _lib.py      ctypes binding + in-tree hipcc build
  api.py       host-side mirror of the reference's interface for this path
  sharded.py   row-sharded multi-process driver (torch.distributed + RCCL)
"""
from . import _lib
from ._lib import build, declared_symbols, needs_build, SO_PATH  # noqa: F401
from .api import *  # noqa: F401,F403
from .api import (CDOptions, IterLassoOptions, ProxL1, SparseIterate, CDLeastSquaresLoss,  # noqa: F401
                  CDSqrtLassoLoss, CDWeightedLSLoss, CoordinateDifferentiableFunction,
                  OrderedIterator, RandomIterator, reset_, numCoordinates, initialize_, gradient,
                  descendCoordinate_, coordinateDescent_, cdPass_, findLambdaMax, stdX, objective,
                  lasso, sqrtLasso, scaledLasso_, LassoPath, LassoSolution, LassoPathResult,
                  DimensionMismatch, ArgumentError, DomainError, HipError)
