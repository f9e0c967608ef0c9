"""pathplanning_amd -- MI355X-native planner core behind the lfilipozzi/PathPlanning plugin surface.

Only the hot path lives here: csrc/ holds the HIP kernels and the
C ABI (include/pp_hip.h); planner.py mirrors the reference's operator interface on top of it.
"""
from . import _lib  # noqa: F401
from ._lib import PPError  # noqa: F401
from .planner import (Context, OccupancyMapSet, StateValidatorOccupancyMap, HybridAStarBatch, ReedsSheppSolver,  # noqa: F401
                      ObstaclesHeuristic, NonHolonomicHeuristic, ReedsSheppPaths, Tree, Status, HybridAStarSearchParameters, HybridAStarPipeline, RRT, RRTStar, GridAStarBatch)
