"""Kinematic tasks (``pink/tasks``)."""
from .frame_task import FrameTask
from .linear_holonomic_task import JointCouplingTask, JointVelocityTask, LinearHolonomicTask
from .posture_task import DampingTask, LowAccelerationTask, PostureTask
from .relative_frame_task import RelativeFrameTask
from .task import Task

__all__ = ["Task", "FrameTask", "RelativeFrameTask", "PostureTask", "DampingTask", "LowAccelerationTask",
           "LinearHolonomicTask", "JointCouplingTask", "JointVelocityTask"]
