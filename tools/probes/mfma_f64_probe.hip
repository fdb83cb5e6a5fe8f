// mfma_f64_probe.hip — derives the lane layouts of the two fp64 MFMA forms on gfx950 by experiment.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// A lane l holds a_l = 1 + l; B is one-hot on lane t (t = 0..63, one MFMA each).  Output lane o, register r,
// then holds a_l for exactly those (o, t) where the product A[lane l] * B[lane t] enters D[o][r]: the table
// of (l, t) -> (o, r) is the layout.  Also times a dependent and an independent chain of each form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe_4x4x4(double* out) {  // out[t][lane]
  const int lane = threadIdx.x;
  const double a = 1.0 + lane;
  for (int t = 0; t < 64; ++t) {
    const double b = lane == t ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[t * 64 + lane] = d;
  }
}
__global__ void probe_16x16x4(double* out) {  // out[t][reg][lane]
  const int lane = threadIdx.x;
  const double a = 1.0 + lane;
  for (int t = 0; t < 64; ++t) {
    const double b = lane == t ? 1.0 : 0.0;
    d4 c = {0, 0, 0, 0};
    const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(t * 4 + r) * 64 + lane] = d[r];
  }
}

template <int FORM, bool DEP>
__global__ void rate(double* out, int iters) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  if (FORM == 0) {
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      if (DEP) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      else c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      if (DEP) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      else c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      if (DEP) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      else c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
  } else if (FORM == 1) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      if (DEP) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      else c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      if (DEP) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      else c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      if (DEP) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      else c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {  // vector FMA chain, 4 per iteration
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_fma(a, b, c0);
      if (DEP) c0 = __builtin_fma(a, b, c0); else c1 = __builtin_fma(a, b, c1);
      if (DEP) c0 = __builtin_fma(a, b, c0); else c2 = __builtin_fma(a, b, c2);
      if (DEP) c0 = __builtin_fma(a, b, c0); else c3 = __builtin_fma(a, b, c3);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
  }
}

template <int FORM, bool DEP>
static void time_rate(const char* name, double flop_per_instr) {
  double* d;
  hipMalloc(&d, sizeof(double) * 256 * 4 * 64);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // one wave per SIMD on every CU: 256 CUs x 4 waves
  hipLaunchKernelGGL((rate<FORM, DEP>), dim3(256), dim3(256), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate<FORM, DEP>), dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = 4.0 * iters;  // per wave
  const double ns_per = ms * 1e6 / instr;
  printf("%-44s %7.2f ns/instr/wave (~%5.1f cycles @2.4GHz)  chip %.1f TFLOP/s\n", name, ns_per, ns_per * 2.4,
         flop_per_instr * instr * 1024 / (ms * 1e-3) / 1e12);
  hipFree(d);
}

int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 64 * 4 * 64);
  std::vector<double> h(64 * 4 * 64);
  hipLaunchKernelGGL(probe_4x4x4, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h.data(), d, sizeof(double) * 64 * 64, hipMemcpyDeviceToHost);
  printf("v_mfma_f64_4x4x4f64: output lane o <- sum over (A lane l, B lane t):\n");
  for (int o = 0; o < 64; ++o) {
    printf("  D lane %2d:", o);
    for (int t = 0; t < 64; ++t)
      if (h[t * 64 + o] != 0.0) printf(" A%02d*B%02d", (int)h[t * 64 + o] - 1, t);
    printf("\n");
  }
  hipLaunchKernelGGL(probe_16x16x4, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h.data(), d, sizeof(double) * 64 * 4 * 64, hipMemcpyDeviceToHost);
  printf("v_mfma_f64_16x16x4f64: (output lane o, reg r) <- (A lane l, B lane t), first 20 lanes:\n");
  for (int o = 0; o < 64; ++o)
    for (int r = 0; r < 4; ++r) {
      if (o >= 20 && o < 60) continue;
      printf("  D lane %2d reg %d:", o, r);
      for (int t = 0; t < 64; ++t)
        if (h[(t * 4 + r) * 64 + o] != 0.0) printf(" A%02d*B%02d", (int)h[(t * 4 + r) * 64 + o] - 1, t);
      printf("\n");
    }
  time_rate<0, true>("v_mfma_f64_4x4x4 dependent chain", 512);
  time_rate<0, false>("v_mfma_f64_4x4x4 4 independent accumulators", 512);
  time_rate<1, true>("v_mfma_f64_16x16x4 dependent chain", 2048);
  time_rate<1, false>("v_mfma_f64_16x16x4 4 independent accumulators", 2048);
  time_rate<2, true>("v_fma_f64 dependent chain", 128);
  time_rate<2, false>("v_fma_f64 4 independent accumulators", 128);
  return 0;
}
