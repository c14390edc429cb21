from .schedulers import FlowMatchDiscreteScheduler  # noqa: F401
