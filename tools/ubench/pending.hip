// Does a launch whose workgroups do not all fit on the chip hold up kernels on OTHER streams?
//
// `hog` = workgroups of 256 threads that ask for 33 KB of LDS (four fit per CU: 1 024 slots on the chip) and spin for a given
// time; `brief` = a prep-like kernel (many 256-thread workgroups, no LDS, one store per thread).  brief is timed alone, next to
// hog launches that fit (grid <= slots) and next to hog launches that leave workgroups pending in the dispatcher.
// What it decided (round 5): see profiles/r05_pending_workgroups.txt and DESIGN.md 4.1 (the DP workers are a bounded pool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void hog(long long ticks, int *sink)
{
	extern __shared__ int lds[];
	const long long t0 = wall_clock64();
	int x = 0;
	while (wall_clock64() - t0 < ticks) { ++x; __builtin_amdgcn_s_sleep(8); }
	lds[threadIdx.x] = x;
	if (x == -1) *sink = lds[(threadIdx.x + 1) & 255];
}
__global__ __launch_bounds__(256) void brief(unsigned *out)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	out[i] = (unsigned)i * 2654435761u;
}
int main(int argc, char **argv)
{
	const int n_brief = 200000;                         // workgroups of the brief kernel (51 M threads: k_prep_rows of one round)
	int *sink; hipMalloc((void**)&sink, 4);
	unsigned *out; hipMalloc((void**)&out, (size_t)n_brief * 256 * 4);
	hipStream_t sa[3], sb;
	for (auto &s : sa) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
	hipEvent_t base, b0, b1, a1[3];
	hipEventCreate(&base), hipEventCreate(&b0), hipEventCreate(&b1);
	for (auto &e : a1) hipEventCreate(&e);
	hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
	struct Case { const char *what; int n_hog; int grid; long long ticks; };
	// 100 MHz ticks.  "fits": the hog grids together stay below the 1 024 slots; "pending": they do not
	const Case cases[] = {
		{ "brief alone", 0, 0, 0 },
		{ "1 hog launch, 768 wgs x 20 ms (fits)", 1, 768, 2000000 },
		{ "1 hog launch, 5 000 wgs x 4 ms (pending; ~20 ms in all)", 1, 5000, 400000 },
		{ "2 hog launches, 400 wgs x 20 ms each (fit)", 2, 400, 2000000 },
		{ "2 hog launches, 5 000 wgs x 4 ms each (pending)", 2, 5000, 400000 },
		{ "3 hog launches, 300 wgs x 20 ms each (fit)", 3, 300, 2000000 },
		{ "3 hog launches, 5 000 wgs x 4 ms each (pending)", 3, 5000, 400000 },
		{ "3 hog launches, 1 024 wgs x 20 ms each (pending: 3 x the slots)", 3, 1024, 2000000 },
	};
	for (int rep = 0; rep < 2; ++rep)
		for (const Case &c : cases) {
			hipDeviceSynchronize();
			hipEventRecord(base, sb);
			for (int k = 0; k < c.n_hog; ++k) {
				hipLaunchKernelGGL(hog, dim3(c.grid), dim3(256), 33 * 1024, sa[k], c.ticks, sink);
				hipEventRecord(a1[k], sa[k]);
			}
			hipEventRecord(b0, sb);
			hipLaunchKernelGGL(brief, dim3(n_brief), dim3(256), 0, sb, out);
			hipEventRecord(b1, sb);
			hipDeviceSynchronize();
			if (rep == 0) continue;
			float tb0, tb1, ta = 0;
			hipEventElapsedTime(&tb0, base, b0), hipEventElapsedTime(&tb1, base, b1);
			for (int k = 0; k < c.n_hog; ++k) { float t; hipEventElapsedTime(&t, base, a1[k]); if (t > ta) ta = t; }
			printf("%-66s brief: start %6.2f ms, end %6.2f ms (%.2f ms)   hogs end %6.2f ms\n", c.what, tb0, tb1, tb1 - tb0, ta);
		}
	return 0;
}
