"""Batched equivalents of the user addons defined in the reference's example
scripts, registered under the names those scripts register
(reference: examples/drone_pilot/drone_pilot.py:58-59)."""
from .addons.addon import AddonFactory
from .addons.controllers import Propellor
from .addons.rewards import FellOver


def register():
    AddonFactory.register_addon('propellor', Propellor)
    AddonFactory.register_addon('fell_over', FellOver)


register()
