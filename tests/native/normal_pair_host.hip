// Host build of the device's words -> N(0,1) pair map (openmcmc_amd/csrc/omc_common.h, omc_normal_pair),
// so that the CPU test suite can hold it to the NumPy model in tests/philox_model.py without a GPU.
//   hipcc -x hip --cuda-host-only -O2 -I include -I openmcmc_amd/csrc tests/native/normal_pair_host.hip -o <exe>
// stdin: N x 4 uint32 words;  stdout: N x 2 float64.
#include <stdio.h>
#include <vector>

#include "omc_common.h"

int main() {
  std::vector<uint32_t> w;
  uint32_t buf[4096];
  size_t got;
  while ((got = fread(buf, 4, 4096, stdin)) > 0) w.insert(w.end(), buf, buf + got);
  const size_t n = w.size() / 4;
  std::vector<double> out(2 * n);
  for (size_t i = 0; i < n; ++i) omc_normal_pair(make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]), out[2 * i], out[2 * i + 1]);
  fwrite(out.data(), 8, out.size(), stdout);
  return 0;
}
