"""CPU oracle: TEST INFRASTRUCTURE ONLY.

A plain NumPy/SciPy restatement of the reference algorithm (sede-open/openMCMC v1.0.7) for the
sampler hot path, with every random draw passed in explicitly ("injected draws") so that the
same numbers can be replayed through the HIP path.  Each function cites the reference
file:line it follows (paths relative to /root/reference/src/openmcmc/).

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which were produced by running the reference itself in the build
container (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under openmcmc_amd/ imports it; the product path has no CPU fallback.
"""
