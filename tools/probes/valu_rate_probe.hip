// valu_rate_probe.hip — issue rate of the VALU instructions the CAAR kernels are made of, per SIMD, at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_rate_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
// Every kernel runs ITER iterations of 8 independent instructions of one kind per wave; one workgroup per CU.
#include <hip/hip_runtime.h>

#include <cstdio>

enum { FMA64, MUL64, ADD64, RCP64, DPP32, MOV32, FMA32, BPERM, MIX };

template <int OP>
__global__ void rate(double* out, int iters) {
  double a[8];
  const double b = 1.0000001, c = 1e-9;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
  int ia[8];
  float fa[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ia[i] = threadIdx.x + i; fa[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
      if (OP == DPP32) asm volatile("v_mov_b32_dpp %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(ia[i]));
      if (OP == MOV32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));
      if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fa[i]) : "v"(fa[(i + 1) & 7]));
      if (OP == BPERM) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));
      if (OP == MIX) {  // one contraction step as the kernels do it: 2 DPP moves + 1 FMA
        if (i & 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        else asm volatile("v_mov_b32_dpp %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(ia[i]));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + ia[i] + fa[i];
  if (s == 123.456) out[0] = s;
}

template <int OP>
void run(const char* what, int cus, double ghz) {
  double* d;
  hipMalloc(&d, 8);
  printf("%-44s", what);
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int threads = 256 * wps, iters = 20000;
    hipLaunchKernelGGL((rate<OP>), dim3(cus), dim3(threads), 0, 0, d, 100);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<OP>), dim3(cus), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = double(iters) * 8 * wps;
    printf("  %d w/SIMD: %5.2f cyc/instr", wps, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
  }
  printf("\n");
  hipFree(d);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate * 1e-6;
  printf("%s: %d CUs, %.2f GHz (cycles below assume that clock)\n", p.name, p.multiProcessorCount, ghz);
  const int cus = p.multiProcessorCount;
  run<FMA64>("v_fma_f64", cus, ghz);
  run<MUL64>("v_mul_f64", cus, ghz);
  run<ADD64>("v_add_f64", cus, ghz);
  run<RCP64>("v_rcp_f64", cus, ghz);
  run<DPP32>("v_mov_b32_dpp row_ror", cus, ghz);
  run<MOV32>("v_add_u32", cus, ghz);
  run<FMA32>("v_fma_f32", cus, ghz);
  run<BPERM>("ds_bpermute_b32 (+wait)", cus, ghz);
  run<MIX>("alternating v_mov_b32_dpp / v_fma_f64", cus, ghz);
  return 0;
}
