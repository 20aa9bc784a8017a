from .glow import ListGlow, GlowStep  # noqa: F401
from .glow_modules import (ActNorm, Conv2dZeros, Conv2dNorm, InvConv, AffineCoupling, Squeeze2d, Split2d,  # noqa: F401
                           BatchNormFlow)
