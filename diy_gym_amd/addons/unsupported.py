"""Registry names that resolve but are outside this round's hot-path scope
(SURVEY.md 2 rows 9, 12, 13, 17, 19-22 and 8(f)).  Constructing one raises
``NotImplementedError`` with the reason, instead of silently doing nothing."""
from .addon import Addon


def _stub(name, why):
    def __init__(self, parent, config):
        raise NotImplementedError("addon '%s' is not implemented in the MI355X backend: %s" % (name, why))

    return type(name, (Addon, ), {'__init__': __init__, '__doc__': why})


StuckJointCost = _stub('stuck_joint_cost', 'the reference implementation raises NameError on first use '
                       '(stuck_joint_cost.py:16-21); there is no behaviour to match')
DrawCoords = _stub('draw_coords', 'GUI-only debug drawing')
