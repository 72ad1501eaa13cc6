"""Physical constants and component indices (values as in reference fdtd/constants.py:5-23)."""

from math import pi

c: float = 299792458.0  # speed of light [m/s]
X, Y, Z = 0, 1, 2  # component / axis indices
mu0: float = 4e-7 * pi  # vacuum permeability
eps0: float = 1.0 / (mu0 * c ** 2)  # vacuum permittivity
eta0: float = mu0 * c  # free-space impedance
