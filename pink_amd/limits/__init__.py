"""Kinematic limits (``pink/limits``)."""
from .configuration_limit import ConfigurationLimit
from .limit import Limit
from .velocity_limit import VelocityLimit

__all__ = ["Limit", "ConfigurationLimit", "VelocityLimit"]
