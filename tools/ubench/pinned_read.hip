// How fast does the host read hipHostMalloc'd memory?  (random 4-byte reads and a streaming sum, vs malloc)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const size_t n = 16 << 20;   // 64 MB of int32
	int32_t *pin = nullptr, *pin2 = nullptr;
	hipHostMalloc((void**)&pin, n * 4, hipHostMallocDefault);
	hipHostMalloc((void**)&pin2, n * 4, hipHostMallocNonCoherent);
	int32_t *dev = nullptr;
	hipMalloc((void**)&dev, n * 4);
	hipMemset(dev, 1, n * 4);
	hipMemcpy(pin, dev, n * 4, hipMemcpyDeviceToHost);
	hipMemcpy(pin2, dev, n * 4, hipMemcpyDeviceToHost);
	std::vector<int32_t> mal(n);
	memcpy(mal.data(), pin, n * 4);
	std::vector<uint32_t> idx(1 << 20);
	uint32_t s = 12345;
	for (auto &x : idx) { s = s * 1664525u + 1013904223u; x = s % n; }
	auto bench = [&](const char *name, const int32_t *p) {
		double t0 = now(); long long acc = 0;
		for (size_t i = 0; i < n; ++i) acc += p[i];
		double t1 = now();
		for (uint32_t x : idx) acc += p[x];
		double t2 = now();
		printf("%-28s stream %.2f GB/s, random %.1f ns/read (%lld)\n", name, n * 4 / (t1 - t0) / 1e9, (t2 - t1) / idx.size() * 1e9, acc);
	};
	bench("malloc", mal.data());
	bench("hipHostMalloc default", pin);
	bench("hipHostMalloc noncoherent", pin2);
	return 0;
}
