from .agent import Agent, TreeAgent  # noqa: F401
from .ValueSim import ValueSim  # noqa: F401
from .ValueSimLP import ValueSimLP  # noqa: F401
from .ValueSimC import ValueSimC  # noqa: F401
from .Vanilla import Vanilla  # noqa: F401
from .VanillaC import VanillaC  # noqa: F401
from .DistValueSim import DistValueSim  # noqa: F401
