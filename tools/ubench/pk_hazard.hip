// Does gfx950 need a wait state between a packed (VOP3P) instruction and a VALU instruction that reads its result?  hipcc inserts
// `s_nop 0` there (LLVM's dst_sel forwarding hazard; VOP3P's default op_sel_hi bit shares the bit of VOP3's dst op_sel).  This
// runs a long chain of dependent v_pk_{add,sub}_i16 clamp / v_pk_max_i16 / v_pk_min_i16 with and without the nop, with 1..8
// waves per SIMD, and compares the results lane by lane -- a stale forward would show as a difference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define STEP(NOP) \
	"v_pk_add_i16 %0, %0, %1 clamp\n\t" NOP "v_pk_max_i16 %0, %0, %2\n\t" NOP "v_pk_sub_i16 %0, %0, %3 clamp\n\t" NOP \
	"v_pk_min_i16 %0, %0, %4\n\t" NOP "v_pk_add_i16 %0, %0, %0 clamp\n\t" NOP "v_pk_sub_i16 %0, %0, %1 clamp\n\t" NOP \
	"v_xor_b32 %0, %0, %5\n\t" "v_pk_max_i16 %0, %0, %3\n\t" NOP "v_add_u32 %0, %0, %2\n\t"
template<int NOP> __global__ __launch_bounds__(64) void k_chain(uint32_t *out, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e)
{
	const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
	uint32_t x = gid * 2654435761u;
	const uint32_t va = a + gid, vb = b ^ gid, vc = c + 3 * gid, vd = d - gid, ve = e + (gid << 7);
	for (int it = 0; it < 4000; ++it) {
		if (NOP) asm volatile(STEP("s_nop 0\n\t") STEP("s_nop 0\n\t") STEP("s_nop 0\n\t") STEP("s_nop 0\n\t") : "+v"(x) : "v"(va), "v"(vb), "v"(vc), "v"(vd), "v"(ve));
		else asm volatile(STEP("") STEP("") STEP("") STEP("") : "+v"(x) : "v"(va), "v"(vb), "v"(vc), "v"(vd), "v"(ve));
	}
	out[gid] = x;
}
int main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int n_simd = p.multiProcessorCount * 4;
	uint32_t *o0, *o1;
	hipMalloc(&o0, (size_t)n_simd * 8 * 64 * 4), hipMalloc(&o1, (size_t)n_simd * 8 * 64 * 4);
	for (int W = 1; W <= 8; W <<= 1) {
		const int grid = n_simd * W;
		std::vector<uint32_t> h0((size_t)grid * 64), h1((size_t)grid * 64);
		k_chain<1><<<grid, 64>>>(o1, 0x12345678u, 0x00070003u, 0xfff0ff85u, 0x7f007e00u, 0x01010101u);
		k_chain<0><<<grid, 64>>>(o0, 0x12345678u, 0x00070003u, 0xfff0ff85u, 0x7f007e00u, 0x01010101u);
		hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost);
		hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost);
		size_t diff = 0, zero = 0;
		for (size_t k = 0; k < h0.size(); ++k) diff += h0[k] != h1[k], zero += h0[k] == 0;
		printf("W=%d waves/SIMD: %zu lanes, %zu differ between the chain with and without s_nop 0 (%zu results are zero; sample %08x %08x)\n", W, h0.size(), diff, zero, h0[1], h0[77]);
	}
	return 0;
}
