// Hardware probe (gfx950): semantics of ds_read_b64_tr_b16 and the MFMA C/D layouts used by the kernels.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr.hip -o tools/probe_tr ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_tr(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int l = threadIdx.x;
  // mode 0: lane l reads 8 B at byte offset l*8 (contiguous).  mode 1: row-major [key][16 cols] subtile:
  // lane -> row (l>>2)&3 + 4*(l>>4), col chunk (l&3)*4  (guess at the natural addressing), row stride 32 B
  unsigned addr;
  if (mode == 0) addr = l * 8;
  else addr = (((l >> 2) & 3) + 4 * (l >> 4)) * 32 + (l & 3) * 8;
  unsigned base = (unsigned)(uintptr_t)lds;   // LDS address (low 32 bits of the generic pointer are the LDS offset)
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + base) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}

__global__ void k_mfma(float* c16, float* c32) {
  int l = threadIdx.x;
  // A[i][k] = i (row id), B[k][j] = (k==0) ? j+1000*... : 0  -> C[i][j] = A[i][0]*B[0][j]
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)0.f; b[j] = (__bf16)0.f; }
  // 16x16x32: lane holds A[row=l&15][k=8*(l>>4)+j]; put A[row][k=0] = row+1 ; B[k=0][col] = 1 + col*0.0625
  if ((l >> 4) == 0) { a[0] = (__bf16)(float)((l & 15) + 1); b[0] = (__bf16)(1.f + (l & 15) * 0.0625f); }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) c16[l * 4 + r] = acc[r];
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)0.f; b[j] = (__bf16)0.f; }
  if ((l >> 5) == 0) { a[0] = (__bf16)(float)((l & 31) + 1); b[0] = (__bf16)(1.f + (l & 31) * 0.03125f); }
  f32x16 acc2;
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
  for (int r = 0; r < 16; ++r) c32[l * 16 + r] = acc2[r];
}

int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16 mode %d: lane -> 4 x u16 element indices\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  float *c16, *c32; hipMalloc(&c16, 256 * 4); hipMalloc(&c32, 1024 * 4);
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, c16, c32);
  float h16[256], h32[1024];
  hipMemcpy(h16, c16, sizeof(h16), hipMemcpyDeviceToHost);
  hipMemcpy(h32, c32, sizeof(h32), hipMemcpyDeviceToHost);
  // decode: value = (row+1) * (1 + col/16)  -> check the documented map row = 4*(l>>4)+r, col = l&15
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    float want = (4 * (l >> 4) + r + 1) * (1.f + (l & 15) * 0.0625f);
    if (fabsf(h16[l * 4 + r] - want) > 0.02f * want) ++bad;
  }
  printf("mfma 16x16x32 C layout (row=4*(l>>4)+r, col=l&15): %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
  bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    float want = ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5) + 1) * (1.f + (l & 31) * 0.03125f);
    if (fabsf(h32[l * 16 + r] - want) > 0.02f * want) ++bad;
  }
  printf("mfma 32x32x16 C layout (row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31): %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
  return 0;
}
