// micro-test: operand / result lane maps of v_mfma_f32_16x16x32_f16 and v_mfma_f32_16x16x16_f16 on gfx950, checked with exact integer data
// against a host product (asymmetric operands).  Assumed maps (as documented for the bf16 forms):
//   16x16x32: lane l holds A[row l&15][k = 8 (l>>4) + j], B[k = 8 (l>>4) + j][col l&15], j = 0..7;  16x16x16: k = 4 (l>>4) + j, j = 0..3
//   D: col = l&15, row = 4 (l>>4) + reg
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k32(const int* A, const int* B, float* D)   // A 16x32, B 32x16 row-major ints
{
  const int l = threadIdx.x, c = l & 15, g = l >> 4;
  h8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (_Float16)A[c * 32 + 8 * g + j]; b[j] = (_Float16)B[(8 * g + j) * 16 + c]; }
  f4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(4 * g + r) * 16 + c] = d[r];
}
__global__ void k16(const int* A, const int* B, float* D)   // A 16x16, B 16x16
{
  const int l = threadIdx.x, c = l & 15, g = l >> 4;
  h4 a, b;
  for (int j = 0; j < 4; j++) { a[j] = (_Float16)A[c * 16 + 4 * g + j]; b[j] = (_Float16)B[(4 * g + j) * 16 + c]; }
  f4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(4 * g + r) * 16 + c] = d[r];
}
int main()
{
  int hA[16 * 32], hB[32 * 16]; float hD[256];
  for (int i = 0; i < 16 * 32; i++) { hA[i] = (i * 7 + 3) % 41 - 20; hB[i] = (i * 13 + 5) % 37 - 18; }
  int *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int K = 32; K >= 16; K -= 16)
  {
    if (K == 32) hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD); else hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++)
    {
      long s = 0; for (int k = 0; k < K; k++) s += (long)hA[m * K + k] * hB[k * 16 + n];
      if ((long)hD[m * 16 + n] != s) bad++;
    }
    printf("mfma_f32_16x16x%d_f16 lane maps: %s (%d mismatches)\n", K, bad ? "WRONG" : "as assumed", bad);
  }
  return 0;
}
