// Host build of the device's Phi^-1 (openmcmc_amd/csrc/omc_truncnorm.h: omc_ndtri_as241) for the CPU test suite.
//   hipcc -x hip --cuda-host-only -O2 -I include -I openmcmc_amd/csrc tests/native/ndtri_host.hip -o <exe>
// stdin: float64 probabilities;  stdout: float64 quantiles.
#include <stdio.h>
#include <vector>

#include "omc_common.h"
#include "omc_truncnorm.h"

int main() {
  std::vector<double> p;
  double buf[4096];
  size_t got;
  while ((got = fread(buf, 8, 4096, stdin)) > 0) p.insert(p.end(), buf, buf + got);
  for (auto& v : p) v = omc_ndtri_as241(v);
  fwrite(p.data(), 8, p.size(), stdout);
  return 0;
}
