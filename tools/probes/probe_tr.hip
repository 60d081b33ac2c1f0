// Probe of ds_read_b64_tr_b16 semantics on gfx950 (build: hipcc --offload-arch=gfx950 probe_tr.hip -o probe_tr).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const short* in, short* out, const int* addr) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = in[i];
  __syncthreads();
  auto p = (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  *(s16x4*)(out + threadIdx.x * 4) = v;
}
int main() {
  std::vector<short> in(64 * 64), out(256);
  for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) in[r * 64 + c] = (short)(r * 64 + c);
  std::vector<int> addr(64);
  for (int l = 0; l < 64; ++l) { int g = l / 16, i = l % 16, q = i / 4, p = i % 4; addr[l] = ((8 * g + q) * 64 + 4 * p) * 2; }
  short *din, *dout; int* daddr;
  hipMalloc(&din, in.size() * 2); hipMalloc(&dout, 512); hipMalloc(&daddr, 256);
  hipMemcpy(din, in.data(), in.size() * 2, hipMemcpyHostToDevice); hipMemcpy(daddr, addr.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, daddr);
  hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d (addr row %d col %d):", l, addr[l] / 128, (addr[l] % 128) / 2);
    for (int e = 0; e < 4; ++e) printf("  (r%d,c%d)", out[l * 4 + e] / 64, out[l * 4 + e] % 64);
    printf("\n");
  }
  return 0;
}
