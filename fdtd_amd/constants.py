"""Axis indices and SI constants used by the host code.

Same names and values as the reference exposes (reference fdtd/constants.py:5-23), so
user scripts that import them keep working.
"""

import math

# component / axis indices of the trailing dimension of the field arrays
X, Y, Z = range(3)

# vacuum: speed of light [m/s], permeability [H/m], then the two derived quantities
c = 299_792_458.0
mu0 = 4e-7 * math.pi
eps0 = 1.0 / (mu0 * c ** 2)   # permittivity [F/m]
eta0 = mu0 * c                # wave impedance [ohm]
