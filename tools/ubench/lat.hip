// Single-wave latency microbenchmarks for gfx950: what does ONE wavefront pay per instruction?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 256
__global__ void k_dep_pk(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t b) {
  uint32_t x = a + threadIdx.x;
  uint64_t t0 = clock64();
  for (int it = 0; it < 1000; ++it) {
#pragma unroll
    for (int k = 0; k < REP; ++k) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(b));
  }
  uint64_t t1 = clock64();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_indep4_pk(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t b) {
  uint32_t x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  uint64_t t0 = clock64();
  for (int it = 0; it < 1000; ++it) {
#pragma unroll
    for (int k = 0; k < REP / 4; ++k) asm volatile("v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));
  }
  uint64_t t1 = clock64();
  out[threadIdx.x] = x0 ^ x1 ^ x2 ^ x3; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dep_add32(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t b) {
  uint32_t x = a + threadIdx.x;
  uint64_t t0 = clock64();
  for (int it = 0; it < 1000; ++it) {
#pragma unroll
    for (int k = 0; k < REP; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
  }
  uint64_t t1 = clock64();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dep_dpp(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t b) {
  uint32_t x = a + threadIdx.x, t = 0;
  uint64_t t0 = clock64();
  for (int it = 0; it < 1000; ++it) {
#pragma unroll
    for (int k = 0; k < REP / 2; ++k) asm volatile("s_nop 1\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_pk_max_i16 %0, %0, %1" : "+v"(x), "+v"(t));
  }
  uint64_t t1 = clock64();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds_chase(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t b) {
  __shared__ uint32_t tab[1024];
  for (int k = threadIdx.x; k < 1024; k += 64) tab[k] = ((k + 64) & 1023) * 4;
  __syncthreads();
  uint32_t p = threadIdx.x * 4;
  uint64_t t0 = clock64();
  for (int it = 0; it < 1000; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(p));
  }
  uint64_t t1 = clock64();
  out[threadIdx.x] = p; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_salu(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t b) {
  uint32_t x = a;
  uint64_t t0 = clock64();
  for (int it = 0; it < 1000; ++it) {
#pragma unroll
    for (int k = 0; k < REP; ++k) asm volatile("s_add_u32 %0, %0, %1" : "+s"(x) : "s"(b));
  }
  uint64_t t1 = clock64();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_ballot_branch(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t b) {
  uint32_t x = a + threadIdx.x, acc = 0;
  uint64_t t0 = clock64();
  for (int it = 0; it < 100000; ++it) {
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (__ballot(x == 0xdeadbeef) != 0) acc += 7;       // never taken
  }
  uint64_t t1 = clock64();
  out[threadIdx.x] = x + acc; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  uint32_t *out; uint64_t *cyc; hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
  uint64_t h; 
#define RUN(name, nops, blocks) do { for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL(name, dim3(blocks), dim3(64), 0, 0, out, cyc, 3u, 5u); hipDeviceSynchronize(); } hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-18s blocks=%-5d %8.2f clk/op\n", #name, blocks, (double)h / (nops)); } while (0)
  for (int blocks : {1, 1024, 4096}) {
    RUN(k_dep_pk, 1000.0 * REP, blocks);
    RUN(k_indep4_pk, 1000.0 * REP, blocks);
    RUN(k_dep_add32, 1000.0 * REP, blocks);
    RUN(k_dep_dpp, 1000.0 * REP / 2, blocks);
    RUN(k_lds_chase, 1000.0 * 16, blocks);
    RUN(k_salu, 1000.0 * REP, blocks);
    RUN(k_ballot_branch, 100000.0, blocks);
  }
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); printf("clock kHz %d; clock64 = s_memtime ticks\n", clk);
  int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0); printf("wall clock kHz %d\n", wc);
  return 0;
}
