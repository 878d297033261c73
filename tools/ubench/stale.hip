// Can a workgroup that is ALREADY RUNNING read stale data that another stream rewrote through kernel boundaries?
//
// `reader`: one persistent workgroup per CU slot (grid 1024 x 64 threads), no fence of its own.  Round r: wait until the flag says r
// (agent-scope atomic load), re-read the SAME n words it read in every round before (plain loads: they sit in its CU's L1 and
// its XCD's L2) and count the words that are not r.  Host, round r: rewrite the words with r -- by a fill kernel, or by a
// host-to-device copy -- then a one-thread kernel that stores the flag (both on another stream, in stream order).
// The question behind MPA_DP_ACQUIRE (dp_kernels.hip, k_dp_worker): do the kernel boundaries between a round's producers and its
// arming make the round's inputs visible to workers that were launched earlier, or does each worker need its own acquire?
//   stale [words per workgroup] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void reader(const int *flag, const int *data, int n_words, int rounds, int mode, unsigned long long *stale, int *ack)
{
	const int *mine = data + (size_t)blockIdx.x * n_words;
	unsigned long long bad = 0;
	for (int r = 1; r <= rounds; ++r) {
		while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r) __builtin_amdgcn_s_sleep(16);
		if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		asm volatile("" ::: "memory");
		for (int k = threadIdx.x; k < n_words; k += 64) bad += mine[k] != r;
		asm volatile("" ::: "memory");
		__syncthreads();
		if (threadIdx.x == 0) __hip_atomic_fetch_add(ack, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	if (bad) atomicAdd(stale, bad);
}
__global__ void fill(int *data, size_t n, int v) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) data[i] = v; }
__global__ void set_flag(int *flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main(int argc, char **argv)
{
	const int n_words = argc > 1 ? atoi(argv[1]) : 1024, rounds = argc > 2 ? atoi(argv[2]) : 200, n_wg = 1024;
	const size_t n = (size_t)n_wg * n_words;
	int *flag, *data, *ack, *h_src;
	unsigned long long *stale;
	hipMalloc((void**)&flag, 4), hipMalloc((void**)&ack, 4), hipMalloc((void**)&stale, 8), hipMalloc((void**)&data, n * 4);
	hipHostMalloc((void**)&h_src, n * 4, hipHostMallocDefault);
	hipStream_t sr, sw;
	hipStreamCreateWithFlags(&sr, hipStreamNonBlocking), hipStreamCreateWithFlags(&sw, hipStreamNonBlocking);
	for (int producer = 0; producer < 2; ++producer)          // 0: fill kernel, 1: host-to-device copy
		for (int mode = 0; mode < 2; ++mode) {                  // 0: the reader has no fence, 1: agent-scope acquire per round
			hipMemset(flag, 0, 4), hipMemset(ack, 0, 4), hipMemset(stale, 0, 8), hipMemset(data, 0, n * 4);
			hipDeviceSynchronize();
			hipLaunchKernelGGL(reader, dim3(n_wg), dim3(64), 0, sr, flag, data, n_words, rounds, mode, stale, ack);
			for (int r = 1; r <= rounds; ++r) {
				if (producer == 0) hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sw, data, n, r);
				else { for (size_t i = 0; i < n; ++i) h_src[i] = r; hipMemcpyAsync(data, h_src, n * 4, hipMemcpyHostToDevice, sw); }
				hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, sw, flag, r);
				hipStreamSynchronize(sw);
				int a = 0;                                          // every reader has finished the round before the words change again
				for (int spin = 0; spin < 200000; ++spin) { hipMemcpy(&a, ack, 4, hipMemcpyDeviceToHost); if (a >= n_wg * r) break; }
				if (a < n_wg * r) { printf("reader stuck at round %d (%d acks)\n", r, a); return 1; }
			}
			hipStreamSynchronize(sr);
			unsigned long long s = 0;
			hipMemcpy(&s, stale, 8, hipMemcpyDeviceToHost);
			printf("%d words x %d workgroups, %d rounds, producer %s, reader %s: %llu stale words of %llu\n", n_words, n_wg, rounds, producer ? "H2D copy" : "fill kernel",
			       mode ? "agent acquire" : "no fence", s, (unsigned long long)n * rounds);
		}
	return 0;
}
