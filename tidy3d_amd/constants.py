"""Physical constants in tidy3d's unit system (micrometres, seconds, Hz).

The numerical values are physical facts; the unit system (um) follows
reference tidy3d/constants.py:16-32 so that every quantity a tidy3d.Simulation
carries can be used without conversion.
"""
import numpy as np

C_0 = 2.99792458e14                    # speed of light [um/s]     (ref constants.py:16)
MU_0 = 1.25663706212e-12               # vacuum permeability [H/um] (ref constants.py:21)
EPSILON_0 = 1.0 / (MU_0 * C_0 ** 2)    # vacuum permittivity [F/um] (ref constants.py:26)
ETA_0 = float(np.sqrt(MU_0 / EPSILON_0))  # vacuum impedance [Ohm]  (ref constants.py:32)

fp_eps = float(np.finfo(np.float32).eps)   # ref constants.py:58
dp_eps = float(np.finfo(np.float64).eps)
inf = float("inf")
LARGE_NUMBER = 1e10

# spectrum(): amplitudes relatively smaller than this are cut (ref components/time.py:17)
DFT_CUTOFF = 1e-8
# GaussianPulse.end_time factor (ref components/source.py END_TIME_FACTOR_GAUSSIAN)
END_TIME_FACTOR_GAUSSIAN = 10
