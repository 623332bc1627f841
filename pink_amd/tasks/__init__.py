"""Kinematic tasks (``pink/tasks``)."""
from .frame_task import FrameTask
from .posture_task import DampingTask, LowAccelerationTask, PostureTask
from .task import Task

__all__ = ["Task", "FrameTask", "PostureTask", "DampingTask", "LowAccelerationTask"]
