"""Argument checks with the behaviour of the reference's extern/validator.py:8-85
(TypeError for a wrong type / physical type, ValueError for a domain violation),
extended to accept a 1-D batch over walkers where the reference takes a scalar."""
import numpy as np

from . import units as u


def validate_physical_type(name, value, physical_type):
    if physical_type is not None:
        if not isinstance(value, u.Quantity):
            raise TypeError("{0} should be given as a Quantity object".format(name))
        pts = physical_type if isinstance(physical_type, (list, tuple)) else [physical_type]
        if value.unit.physical_type not in pts:
            raise TypeError("{0} should be given in units of {1}".format(name, ", ".join(pts)))


def _check_domain(name, value, domain):
    v = value.value if isinstance(value, u.Quantity) else value
    if getattr(v, "__array_priority__", 0) == 30000:
        return  # device-resident: checking would force a synchronising download
    v = np.asarray(v, dtype=float)
    if np.any(~np.isfinite(v)):
        raise ValueError("{0} value is NaN or Inf".format(name))
    if domain is None:
        return
    if domain == "positive":
        if np.any(v < 0):
            raise ValueError("{0} value should be positive".format(name))
    elif domain == "strictly-positive":
        if np.any(v <= 0):
            raise ValueError("{0} value should be strictly positive".format(name))
    elif domain == "negative":
        if np.any(v > 0):
            raise ValueError("{0} value should be negative".format(name))
    elif isinstance(domain, (tuple, list)) and len(domain) == 2:
        if np.any(v < domain[0]) or np.any(v > domain[-1]):
            raise ValueError("{0} values must be in domain {1}".format(name, domain))


def validate_scalar(name, value, domain=None, physical_type=None):
    validate_physical_type(name, value, physical_type)
    if np.ndim(value.value if isinstance(value, u.Quantity) else value) != 0:
        raise TypeError("{0} should be a scalar".format(name))
    _check_domain(name, value, domain)
    return value


def validate_scalar_or_batch(name, value, domain=None, physical_type=None):
    """a scalar (reference behaviour) or a 1-D array over walkers"""
    validate_physical_type(name, value, physical_type)
    if np.ndim(value.value if isinstance(value, u.Quantity) else value) > 1:
        raise TypeError("{0} should be a scalar or a 1-D batch over walkers".format(name))
    _check_domain(name, value, domain)
    return value


def validate_array(name, value, domain=None, ndim=1, shape=None, physical_type=None):
    validate_physical_type(name, value, physical_type)
    v = value.value if isinstance(value, u.Quantity) else np.asarray(value)
    if ndim is not None and np.ndim(v) != ndim:
        raise TypeError("{0} should be a {1}-dimensional array".format(name, ndim))
    if shape is not None and np.shape(v) != tuple(shape):
        raise ValueError("{0} should be an array of shape {1}".format(name, shape))
    _check_domain(name, value, domain)
    return value
