from .scheduling_flow_match_discrete import FlowMatchDiscreteScheduler  # noqa: F401
