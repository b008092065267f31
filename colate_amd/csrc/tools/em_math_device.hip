// colate_amd/csrc/tools/em_math_device.hip -- runs the kernel's math header (em_math.hpp) ON THE DEVICE over a file
// of doubles, for tests/test_gpu_em_math.py: em_exp / em_exp_om / em_log are the same source on host and device,
// but em_rcp, fma_cc, max_c and ldexp/frexp take device-only code (v_rcp_f64 + Newton, inline v_fma_f64 / v_max_f64,
// v_ldexp_f64 / v_frexp_*), so "identical on gfx950" is checked here bit for bit instead of assumed.
//   em_math_device IN OUT : IN = n doubles (x) followed by n doubles (aux); OUT = 11 arrays of n doubles:
//   em_exp(x) | em_exp_om(x).value | em_exp_om(x).one_minus | em_log(|x|) | em_rcp(|x|) | em_div_known_rcp(aux + 1/|x|, 1/|x|, |x|)
//   | 1/|x| by IEEE division on the device | em_exp_t(x) | em_exp_om_t(x).value | em_exp_om_t(x).one_minus
//   | em_rcp_ieee(|x|)
//   (the table-driven exp the kernel uses, here with the table read from device memory instead of LDS)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../em_math.hpp"

__global__ void probe(int n, const double* __restrict__ x, const double* __restrict__ aux, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i], a = v < 0 ? -v : v;
  double om;
  out[0 * (size_t)n + i] = em::em_exp(v);
  out[1 * (size_t)n + i] = em::em_exp_om(v, &om);
  out[2 * (size_t)n + i] = om;
  out[3 * (size_t)n + i] = em::em_log(a);
  out[4 * (size_t)n + i] = em::em_rcp(a);
  const double inv = 1.0 / a;
  out[5 * (size_t)n + i] = em::em_div_known_rcp(aux[i] + inv, inv, a);
  out[6 * (size_t)n + i] = inv;
  double om_t;
  out[7 * (size_t)n + i] = em::em_exp_t(v, em::kExpTableDevice);
  out[8 * (size_t)n + i] = em::em_exp_om_t(v, &om_t, em::kExpTableDevice);
  out[9 * (size_t)n + i] = om_t;
  out[10 * (size_t)n + i] = em::em_rcp_ieee(a);
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  const int n = (int)(bytes / 16);
  std::vector<double> in(2 * (size_t)n), out(11 * (size_t)n);
  if (fread(in.data(), 8, 2 * (size_t)n, f) != 2 * (size_t)n) return 2;
  fclose(f);
  double *d_in, *d_out;
  if (hipMalloc(&d_in, in.size() * 8) != hipSuccess || hipMalloc(&d_out, out.size() * 8) != hipSuccess) return 3;
  if (hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return 3;
  hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_in, d_in + n, d_out);
  if (hipDeviceSynchronize() != hipSuccess) return 3;
  if (hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
  f = fopen(argv[2], "wb");
  if (!f || fwrite(out.data(), 8, out.size(), f) != out.size()) return 2;
  fclose(f);
  return 0;
}
