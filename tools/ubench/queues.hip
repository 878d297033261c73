// How many kernels on different HIP streams really run at once?  n_streams one-wave kernels of ~20 ms each, launched together;
// prints the makespan and, per stream, when its kernel started (HIP events).  Run under different GPU_MAX_HW_QUEUES.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long long cycles, int *sink)
{
	const long long t0 = wall_clock64();
	int x = 0;
	while (wall_clock64() - t0 < cycles) ++x;
	if (x == -1) *sink = x;
}
int main(int argc, char **argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 24;
	const long long cycles = 2000000;                 // wall_clock64 ticks at 100 MHz -> 20 ms
	std::vector<hipStream_t> st(n);
	std::vector<hipEvent_t> e0(n), e1(n);
	int *sink; hipMalloc((void**)&sink, 4);
	hipEvent_t base; hipEventCreate(&base);
	for (int i = 0; i < n; ++i) { hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); hipEventCreate(&e0[i]); hipEventCreate(&e1[i]); }
	for (int rep = 0; rep < 2; ++rep) {
		hipDeviceSynchronize();
		hipEventRecord(base, st[0]);
		for (int i = 0; i < n; ++i) { hipEventRecord(e0[i], st[i]); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], cycles, sink); hipEventRecord(e1[i], st[i]); }
		hipDeviceSynchronize();
		if (rep == 0) continue;
		float worst = 0;
		printf("GPU_MAX_HW_QUEUES=%s streams=%d: start/end (ms) per stream:", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)", n);
		for (int i = 0; i < n; ++i) {
			float a, b; hipEventElapsedTime(&a, base, e0[i]); hipEventElapsedTime(&b, base, e1[i]);
			if (b > worst) worst = b;
			printf(" %d:%.0f-%.0f", i, a, b);
		}
		printf("\n  makespan %.1f ms (one kernel = 20 ms)\n", worst);
	}
	return 0;
}
