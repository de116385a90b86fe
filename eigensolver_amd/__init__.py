"""eigensolver_amd -- MI355X-native dispersion-relation hot path."""
from ._lib import Context, EsError, load  # noqa: F401
from .slab_analytic import SlabSteadyFlow  # noqa: F401
from . import equilibrium  # noqa: F401
from .shooting import ShootProblem  # noqa: F401
from .solvers import (CylinderNonUniformDensity, CylinderNonUniformFlow, CylinderRotationalFlow,  # noqa: F401
                      SlabNonUniformDensity, SlabNonUniformFlow, SlabUniformFlow)
from .cyl_uniform import CylinderUniform  # noqa: F401
from . import postprocess  # noqa: F401
from .complex_flow import SlabComplexFlow, SlabFlowKH  # noqa: F401
