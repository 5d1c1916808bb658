"""naima_amd -- the radiative-likelihood hot path of naima on AMD MI355X (gfx950).

Drop-in for that path only: the particle distributions and radiative models of
``naima.models``, ``naima.core.lnprob`` and the ``(pars, data) -> (flux, *blobs)``
model-function contract, executed by hand-written HIP kernels
(``libnaima_hip.so``, C ABI in ``include/naima_hip.h``) and vectorised over the
walkers of an ensemble.  ``naima_amd.units`` spells the subset of
``astropy.units`` those model functions use.
"""
from . import units  # noqa: F401
from . import units as u  # noqa: F401
from .core import (lnprob, lnprobmodel, log_uniform_prior, normal_prior,  # noqa: F401
                   uniform_prior, get_sampler, run_sampler)
from .models import (BrokenPowerLaw, ExponentialCutoffBrokenPowerLaw,  # noqa: F401
                     ExponentialCutoffPowerLaw, LogParabola, PowerLaw, TableModel,
                     EblAbsorptionModel)
from .radiative import (Bremsstrahlung, InverseCompton, PionDecay,  # noqa: F401
                        PionDecayKelner06, Synchrotron)
from .analysis import find_ML, read_run, save_results_table, save_run  # noqa: F401
from .datatable import validate_data_table  # noqa: F401

__version__ = "0.1.0"
