"""pink_amd: MI355X-native batched differential inverse kinematics behind Pink's API.

The public surface mirrors ``pink/__init__.py`` (``solve_ik``, ``build_ik``,
``Configuration``, ``Task``) and adds ``solve_ik_batch`` and the packed-batch
interface (``IKBatch``, ``BatchSolver``).
"""

__version__ = "0.1.0"

from .batch import IKBatch, pack_terms
from .batch_solver import BatchResult, BatchSolver
from .configuration import Configuration, ConfigurationBatch, Model, build_chain, load_urdf
from .exceptions import NoSolutionFound, NotWithinConfigurationLimits, PinkError, TargetNotSet
from .sharding import MultiDeviceSolver
from .solve_ik import build_ik, clear_device_cache, last_solve_stats, pack_configurations, pinned_empty, solve_ik, solve_ik_batch
from .tasks import DampingTask, FrameTask, PostureTask, Task

__all__ = [
    "BatchResult", "BatchSolver", "Configuration", "ConfigurationBatch", "DampingTask", "FrameTask", "IKBatch", "Model", "MultiDeviceSolver", "NoSolutionFound",
    "NotWithinConfigurationLimits", "PinkError", "PostureTask", "TargetNotSet", "Task", "build_chain", "build_ik", "clear_device_cache",
    "last_solve_stats", "load_urdf", "pack_configurations", "pinned_empty", "pack_terms", "solve_ik", "solve_ik_batch",
]
