"""openmcmc_amd: MI355X-native sampler core behind openMCMC's sampler/distribution plugin API.

Importing the package does not load the HIP library; `openmcmc_amd.engine` (and everything
that computes) does, and fails loudly if libomcmc_hip.so has not been built.
"""

__version__ = "0.1.0"
