// Issue / throughput probe for gfx950 VALU, transcendental, packed and MFMA instructions, alone and interleaved, at 1 / 2 / 4 waves per SIMD.
// Each wave times a loop of independent instructions with s_memtime (shader cycles); prints cycles per instruction per WAVE and per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_rates.hip -o tools/probes/valu_rates && tools/probes/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void probe(uint64_t* out, int iters, float seed) {
  float a[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f32x2{a[i], a[i] + 0.5f}; }
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, 0.f, 1.f, 2.f};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
  const float c1 = 0.99f, c2 = 1e-3f;
  uint64_t t0 = __builtin_readcyclecounter();
  asm volatile("s_nop 0" ::: "memory");
  uint64_t s0, s1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0) :: "memory");
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c1), "v"(c2)); }
    if (OP == 1) { asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(p[0]), "v"(p[1])); }
#define UN8(OPC) asm volatile(OPC " %0, %0\n " OPC " %1, %1\n " OPC " %2, %2\n " OPC " %3, %3\n " OPC " %4, %4\n " OPC " %5, %5\n " OPC " %6, %6\n " OPC " %7, %7\n" \
                              OPC " %0, %0\n " OPC " %1, %1\n " OPC " %2, %2\n " OPC " %3, %3\n " OPC " %4, %4\n " OPC " %5, %5\n " OPC " %6, %6\n " OPC " %7, %7\n" \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
#define BI8(OPC) asm volatile(OPC " %0, %0, %8\n " OPC " %1, %1, %8\n " OPC " %2, %2, %8\n " OPC " %3, %3, %8\n " OPC " %4, %4, %8\n " OPC " %5, %5, %8\n " OPC " %6, %6, %8\n " OPC " %7, %7, %8\n" \
                              OPC " %0, %0, %8\n " OPC " %1, %1, %8\n " OPC " %2, %2, %8\n " OPC " %3, %3, %8\n " OPC " %4, %4, %8\n " OPC " %5, %5, %8\n " OPC " %6, %6, %8\n " OPC " %7, %7, %8\n" \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c1))
    if (OP == 2) UN8("v_exp_f32");
    if (OP == 3) UN8("v_rcp_f32");
    if (OP == 4) BI8("v_mul_f32");
    if (OP == 5) BI8("v_cvt_pk_bf16_f32");
    if (OP == 6) { asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
                                "v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c1), "v"(c2)); }
    if (OP == 7) {      // 16 MFMAs, 4 independent accumulators
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[k & 3], 0, 0, 0);
    }
    if (OP == 8 || OP == 9 || OP == 10 || OP == 14 || OP == 15) {   // 16 x {1 MFMA + NV VALU}: hand-placed
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[k & 3], 0, 0, 0);
        if (OP == 8) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(c1), "v"(c2));
        if (OP == 9) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c1), "v"(c2));
        if (OP == 10) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(a[0]), "+v"(a[1]));
        if (OP == 14) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(p[4]), "v"(p[5]));
        if (OP == 15) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a[0]), "+v"(a[1]) : "v"(c1), "v"(c2));
      }
    }
    if (OP == 11) { asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                                 "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(p[0])); }
    if (OP == 20) BI8("v_pk_mul_f16");
    if (OP == 21) BI8("v_pk_max_f16");
    if (OP == 22) UN8("v_exp_f16");
    if (OP == 23) UN8("v_rcp_f16");
    if (OP == 24) BI8("v_cvt_pkrtz_f16_f32");
    if (OP == 25) { asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_pk_fma_f16 %7, %7, %8, %9\n"
                                 "v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_pk_fma_f16 %7, %7, %8, %9\n"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c1), "v"(c2)); }
    if (OP == 26 || OP == 27 || OP == 28) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[k & 3], 0, 0, 0);
        if (OP == 26) asm volatile("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5\n" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(c1), "v"(c2));
        if (OP == 27) asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_pk_fma_f16 %7, %7, %8, %9\n"
                                  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c1), "v"(c2));
        if (OP == 28) asm volatile("v_exp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5\n" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(c1), "v"(c2));
      }
    }
    if (OP == 12) { asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n") REP8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a[0]) : "v"(c1), "v"(c2)); }     // dependent chain
    if (OP == 13) { asm volatile(REP8("v_exp_f32 %0, %0\n v_rcp_f32 %0, %0\n") : "+v"(a[0])); }                                         // dependent trans chain (16 instrs)
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1) :: "memory");
  float sink = 0.f;
  for (int i = 0; i < 8; ++i) sink += a[i] + p[i][0] + p[i][1];
  for (int i = 0; i < 4; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s1 - s0;
  if (sink == 12345.678f) out[0] = (uint64_t)t0;
}

template <int OP> void run(const char* name, int ninstr, uint64_t* dbuf) {
  const int iters = 2000;
  for (int W : {1, 2, 3, 4}) {
    const int blocks = 256 * W;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, dbuf, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, dbuf, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * 4);
    hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2], per = med / ((double)iters * ninstr);
    printf("%-34s W=%d  s_memtime ticks/instr/wave %7.2f  -> per SIMD %6.2f   (wall %.1f us: %.2f ns/instr/wave)\n", name, W, per, per / W, ms * 1e3, ms * 1e6 / ((double)iters * ninstr));
  }
}

int main() {
  uint64_t* dbuf; hipMalloc(&dbuf, 8 * 4 * 256 * 8);
  run<0>("v_fma_f32 x16 indep", 16, dbuf);
  run<1>("v_pk_fma_f32 x16 indep", 16, dbuf);
  run<11>("v_pk_mul_f32 x16 indep", 16, dbuf);
  run<4>("v_mul_f32 x16", 16, dbuf);
  run<6>("v_med3_f32 x16", 16, dbuf);
  run<5>("v_cvt_pk_bf16_f32 x16", 16, dbuf);
  run<2>("v_exp_f32 x16 indep", 16, dbuf);
  run<3>("v_rcp_f32 x16 indep", 16, dbuf);
  run<12>("v_fma_f32 x16 dependent chain", 16, dbuf);
  run<13>("v_exp/v_rcp x16 dependent chain", 16, dbuf);
  run<7>("mfma16x16x32 x16 (4 acc)", 16, dbuf);
  run<15>("16 x {mfma + 2 v_fma}  per group", 16, dbuf);
  run<8>("16 x {mfma + 4 v_fma}  per group", 16, dbuf);
  run<9>("16 x {mfma + 8 v_fma}  per group", 16, dbuf);
  run<14>("16 x {mfma + 4 v_pk_fma} per group", 16, dbuf);
  run<10>("16 x {mfma + 2 v_exp}  per group", 16, dbuf);
  run<25>("v_pk_fma_f16 x16 indep", 16, dbuf);
  run<20>("v_pk_mul_f16 x16", 16, dbuf);
  run<21>("v_pk_max_f16 x16", 16, dbuf);
  run<22>("v_exp_f16 x16", 16, dbuf);
  run<23>("v_rcp_f16 x16", 16, dbuf);
  run<24>("v_cvt_pkrtz_f16_f32 x16", 16, dbuf);
  run<26>("16 x {mfma + 4 v_pk_fma_f16}", 16, dbuf);
  run<27>("16 x {mfma + 8 v_pk_fma_f16}", 16, dbuf);
  run<28>("16 x {mfma + exp16 rcp16 2pkfma16}", 16, dbuf);
  return 0;
}
