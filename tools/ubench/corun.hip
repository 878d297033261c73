// Co-run microbenchmark (gfx950): how much does a grid of short, latency-bound waves (the seeding / refinement / preparation kernels'
// shape: dependent loads from a large array, ~64 VGPRs) stretch while long-lived, VALU-bound, register-heavy workgroups (the DP
// round's shape) are resident on every CU -- and does it depend on the registers and the LDS those workgroups hold?
//   A<VG>: 256-thread workgroups that hold VG VGPRs and `lds` bytes of LDS and spin on a dependent packed-int16 chain for ~T ms
//   B:     n_waves one-wave workgroups, each 8 dependent 128-byte-line loads from a 2-GB array, 64 VGPRs
// Reported: B alone, B next to 256 x {2, 3, 4} workgroups of A (one per CU slot), and A's own slowdown.
// Build: hipcc --offload-arch=gfx950 -O2 -o corun corun.hip ; run: ./corun
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template<int VG> __global__ __launch_bounds__(256) void k_long(uint32_t *out, long long ticks, uint32_t a)
{
	extern __shared__ uint32_t lds[];
	constexpr int N = VG - 4;                                  // + 4 of bookkeeping = VG allocated (checked in the ISA's metadata)
	uint32_t x[N];                                             // held in registers for the whole kernel
#pragma unroll
	for (int k = 0; k < N; ++k) x[k] = a + threadIdx.x * (k + 1);
	lds[threadIdx.x] = a;
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) {
#pragma unroll
		for (int r = 0; r < 4; ++r)
#pragma unroll
			for (int k = 0; k < N; ++k) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(x[k]) : "v"(x[(k + 1) % N]));
	}
	uint32_t s = lds[threadIdx.x ^ 1];
#pragma unroll
	for (int k = 0; k < N; ++k) s ^= x[k];
	if (s == 0x12345) out[threadIdx.x] = s;
}

__global__ __launch_bounds__(64) void k_chase(const uint32_t *arr, uint32_t mask, uint32_t *out, int steps, int prio)
{
	if (prio) __builtin_amdgcn_s_setprio(3);                   // (B above A in the SIMD's issue arbitration)
	uint32_t p = (blockIdx.x * 2654435761u + threadIdx.x * 40503u) & mask;
	uint32_t acc = 0, y[56];                                   // 56 + bookkeeping = 64 VGPRs allocated, like the seeding kernels
#pragma unroll
	for (int k = 0; k < 56; ++k) asm volatile("v_mov_b32 %0, %1" : "=v"(y[k]) : "v"(p + k));
	for (int s = 0; s < steps; ++s) { p = arr[p] & mask; acc += p; }   // dependent: one line per step and lane group
#pragma unroll
	for (int k = 0; k < 56; ++k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(y[k]));
	if (acc == 0x12345) out[threadIdx.x] = acc;
}

static int g_b_prio = 0;
static float time_b(hipStream_t sb, const uint32_t *arr, uint32_t mask, uint32_t *out, int n_waves)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipEventRecord(e0, sb);
	k_chase<<<n_waves, 64, 0, sb>>>(arr, mask, out, 8, g_b_prio);
	hipEventRecord(e1, sb);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0), hipEventDestroy(e1);
	return ms;
}

template<int VG> static void run(const char *name, size_t lds, int wg_per_cu, hipStream_t sa, hipStream_t sb, const uint32_t *arr, uint32_t mask, uint32_t *out, int n_waves, float alone)
{
	hipFuncSetAttribute((const void*)k_long<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	const long long ticks = 20 * 100000LL;                      // 20 ms at 100 MHz
	hipEvent_t a0, a1;
	hipEventCreate(&a0), hipEventCreate(&a1);
	hipEventRecord(a0, sa);
	k_long<VG><<<256 * wg_per_cu, 256, lds, sa>>>(out, ticks, 3);
	hipEventRecord(a1, sa);
	// B three times while A is resident
	float b[3];
	for (int r = 0; r < 3; ++r) b[r] = time_b(sb, arr, mask, out, n_waves);
	hipEventSynchronize(a1);
	float ams;
	hipEventElapsedTime(&ams, a0, a1);
	printf("%-44s B %.2f %.2f %.2f ms (alone %.2f: x%.1f); A %.1f ms for a 20-ms spin\n", name, b[0], b[1], b[2], alone, b[1] / alone, ams);
	hipEventDestroy(a0), hipEventDestroy(a1);
}

int main()
{
	const size_t n = (size_t)1 << 29;                           // 2 GB of uint32
	uint32_t *arr, *out;
	hipMalloc(&arr, n * 4), hipMalloc(&out, 4096);
	std::vector<uint32_t> h((size_t)1 << 24);
	for (size_t k = 0; k < h.size(); ++k) h[k] = (uint32_t)(k * 2654435761u + 12345u);
	for (size_t o = 0; o < n; o += h.size()) hipMemcpy(arr + o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	const uint32_t mask = (uint32_t)(n - 1);
	hipStream_t sa, sb;
	hipStreamCreateWithFlags(&sa, hipStreamNonBlocking), hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
	const int n_waves = 400000;
	time_b(sb, arr, mask, out, n_waves);
	float alone = 1e9f;
	for (int r = 0; r < 3; ++r) { const float t = time_b(sb, arr, mask, out, n_waves); alone = t < alone ? t : alone; }
	printf("# B = %d one-wave workgroups x 8 dependent random loads from 2 GB; alone %.2f ms\n", n_waves, alone);
	run<96>("A: 96 VGPRs, 33 KB LDS, 4 workgroups per CU", 33024, 4, sa, sb, arr, mask, out, n_waves, alone);
	run<96>("A: 96 VGPRs, 33 KB LDS, 5 workgroups per CU", 30400, 5, sa, sb, arr, mask, out, n_waves, alone);
	run<124>("A: 124 VGPRs, 41 KB LDS, 3 workgroups per CU", 41216, 3, sa, sb, arr, mask, out, n_waves, alone);
	run<124>("A: 124 VGPRs, 33 KB LDS, 4 workgroups per CU", 33024, 4, sa, sb, arr, mask, out, n_waves, alone);
	run<124>("A: 124 VGPRs, 41 KB LDS, 2 workgroups per CU", 41216, 2, sa, sb, arr, mask, out, n_waves, alone);
	run<124>("A: 124 VGPRs, 41 KB LDS, 1 workgroup per CU", 41216, 1, sa, sb, arr, mask, out, n_waves, alone);
	run<64>("A: 64 VGPRs, 33 KB LDS, 4 workgroups per CU", 33024, 4, sa, sb, arr, mask, out, n_waves, alone);
	// the same with B's waves at s_setprio 3: if B now runs next to A, what held it back was the SIMD's issue arbitration (A's waves are
	// older and each of them can issue a packed add every ~5 cycles: three or four per SIMD leave no slot), not a resource
	g_b_prio = 1;
	printf("# B at s_setprio 3\n");
	run<124>("A: 124 VGPRs, 41 KB LDS, 3 workgroups per CU", 41216, 3, sa, sb, arr, mask, out, n_waves, alone);
	run<124>("A: 124 VGPRs, 41 KB LDS, 2 workgroups per CU", 41216, 2, sa, sb, arr, mask, out, n_waves, alone);
	run<64>("A: 64 VGPRs, 33 KB LDS, 4 workgroups per CU", 33024, 4, sa, sb, arr, mask, out, n_waves, alone);
	return 0;
}
