// probe: relative error of v_rcp_f64 (and of one / two Newton steps on top of it) against IEEE division, gfx950
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
__global__ void k(const double *x, double *e0, double *e1, double *e2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double y = __builtin_amdgcn_rcp(v);
  double ex = 1.0 / v;
  e0[i] = fabs(y - ex) / ex;
  double r = fma(-v, y, 1.0); y = fma(y, r, y);
  e1[i] = fabs(y - ex) / ex;
  r = fma(-v, y, 1.0); y = fma(y, r, y);
  e2[i] = fabs(y - ex) / ex;
}
int main() {
  const int n = 1 << 22;
  double *hx = new double[n], *he = new double[3 * n];
  uint64_t s = 12345;
  for (int i = 0; i < n; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; double u = (double)(s >> 11) / 9007199254740992.0;
    int ex = (int)((s >> 3) % 600) - 300; hx[i] = ldexp(1.0 + u, ex); }
  double *dx, *de;
  hipMalloc(&dx, n * 8); hipMalloc(&de, 3 * n * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, de, de + n, de + 2 * n, n);
  hipMemcpy(he, de, 3 * n * 8, hipMemcpyDeviceToHost);
  for (int j = 0; j < 3; j++) { double m = 0; for (int i = 0; i < n; i++) m = fmax(m, he[j * n + i]); printf("newton steps %d: max rel err %.3e (%.2f ulp)\n", j, m, m / 1.1102230246251565e-16); }
  return 0;
}
