from .addon import Addon, AddonFactory, Receptor  # noqa: F401
