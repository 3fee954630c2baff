"""diy_gym_amd: MI355X-native batched simulation backend behind DIYGym's surface.

``from diy_gym_amd import DIYGym`` mirrors ``from diy_gym import DIYGym``
(reference: diy_gym/__init__.py).
"""
from .config import Configuration  # noqa: F401
from .addons.addon import Addon, AddonFactory, Receptor  # noqa: F401
from .diy_gym import DIYGym  # noqa: F401
from .model import Model  # noqa: F401

__version__ = '0.1.0'
