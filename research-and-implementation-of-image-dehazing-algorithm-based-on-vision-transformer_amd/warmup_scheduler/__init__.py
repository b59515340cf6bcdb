from .scheduler import GradualWarmupScheduler  # noqa: F401
