// VALU issue-rate microbenchmark for gfx950: how many cycles does one SIMD need per wave64 instruction of the kinds the DP row
// sweep is made of, with 1, 2, 4, 8 resident waves per SIMD?  Every wave runs N independent chains of the instruction (no
// dependence stalls inside a wave), the grid fills every SIMD of the chip with W waves, and the kernel is timed from the host:
//   cycles per instruction per SIMD = time * f / (instructions per wave * W)
// f is measured by the same harness with s_memrealtime against clock64 (s_memtime counts shader-clock cycles).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define ITER 2000
#define CHAINS 8
#define REP 8          /* instructions per chain per loop trip */

#define BODY8(INS) \
	asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b) : "vcc", "s20", "s21")

#define I_PKADD(k)  "v_pk_add_i16 %" #k ", %" #k ", %8 clamp\n\t"
#define I_PKADDN(k) "v_pk_add_i16 %" #k ", %" #k ", %8\n\t"
#define I_PKMAX(k)  "v_pk_max_i16 %" #k ", %" #k ", %8\n\t"
#define I_ADD32(k)  "v_add_u32 %" #k ", %" #k ", %8\n\t"
#define I_MAX32(k)  "v_max_i32 %" #k ", %" #k ", %8\n\t"
#define I_FMA(k)    "v_fma_f32 %" #k ", %" #k ", %8, %8\n\t"
#define I_PERM(k)   "v_perm_b32 %" #k ", %" #k ", %8, %8\n\t"
#define I_MOVDPP(k) "v_mov_b32_dpp %" #k ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_MAXDPP(k) "v_max_i32_dpp %" #k ", %8, %" #k " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_SDWA(k)   "v_add_u32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
#define I_MAX3(k)   "v_max3_i32 %" #k ", %" #k ", %8, %8\n\t"
#define I_MOV(k)    "v_mov_b32 %" #k ", %8\n\t"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define I_PKMAD(k)  "v_pk_mad_i16 %" #k ", %" #k ", %8, %8\n\t"
#define I_MED3(k)   "v_med3_i32 %" #k ", %" #k ", %8, %8\n\t"
#define I_MAXI16(k) "v_max_i16 %" #k ", %" #k ", %8\n\t"
#define I_PKMIN(k)  "v_pk_min_i16 %" #k ", %" #k ", %8\n\t"
#define I_PKLSHL(k) "v_pk_lshlrev_b16 %" #k ", 1, %" #k "\n\t"
#define I_SNOP(k)   "s_nop 0\n\t"
#define I_AND(k)    "v_and_b32 %" #k ", %" #k ", %8\n\t"
#define I_OR(k)     "v_or_b32 %" #k ", %" #k ", %8\n\t"
#define I_XOR(k)    "v_xor_b32 %" #k ", %" #k ", %8\n\t"
#define I_LSHL(k)   "v_lshlrev_b32 %" #k ", 1, %" #k "\n\t"
#define I_SUB32(k)  "v_sub_u32 %" #k ", %" #k ", %8\n\t"
#define I_ADD16(k)  "v_add_u16 %" #k ", %" #k ", %8\n\t"
#define I_MINI16(k) "v_min_i16 %" #k ", %" #k ", %8\n\t"
#define I_MAXU32(k) "v_max_u32 %" #k ", %" #k ", %8\n\t"
#define I_MINU32(k) "v_min_u32 %" #k ", %" #k ", %8\n\t"
#define I_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 16, %8\n\t"
#define I_ANDOR(k)  "v_and_or_b32 %" #k ", %" #k ", %8, %8\n\t"
#define I_CMP(k)    "v_cmp_ne_u32 vcc, %" #k ", %8\n\t"
#define I_CND64(k)  "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\n\t"
#define I_ADDI16C(k) "v_add_i16 %" #k ", %" #k ", %8 clamp\n\t"
#define I_BFE(k)    "v_bfe_i32 %" #k ", %" #k ", 0, 16\n\t"
#define I_ADD3(k)   "v_add3_u32 %" #k ", %" #k ", %8, %8\n\t"

#define KERNEL(NAME, INS) \
__global__ __launch_bounds__(64) void NAME(uint32_t *out, uint32_t a, uint32_t b) { \
	uint32_t x[CHAINS]; \
	for (int k = 0; k < CHAINS; ++k) x[k] = a + threadIdx.x * (k + 1); \
	for (int it = 0; it < ITER; ++it) { \
		_Pragma("unroll") for (int r = 0; r < REP; ++r) BODY8(INS); \
	} \
	uint32_t s = 0; for (int k = 0; k < CHAINS; ++k) s ^= x[k]; \
	if (s == 0x12345) out[threadIdx.x] = s; \
}
KERNEL(k_pkadd_clamp, I_PKADD)
KERNEL(k_pkadd, I_PKADDN)
KERNEL(k_pkmax, I_PKMAX)
KERNEL(k_pkmin, I_PKMIN)
KERNEL(k_add32, I_ADD32)
KERNEL(k_max32, I_MAX32)
KERNEL(k_fma32, I_FMA)
KERNEL(k_perm, I_PERM)
KERNEL(k_movdpp, I_MOVDPP)
KERNEL(k_maxdpp, I_MAXDPP)
KERNEL(k_sdwa, I_SDWA)
KERNEL(k_max3, I_MAX3)
KERNEL(k_mov, I_MOV)
KERNEL(k_cndmask, I_CNDMASK)
KERNEL(k_pkmad, I_PKMAD)
KERNEL(k_med3, I_MED3)
KERNEL(k_maxi16, I_MAXI16)
KERNEL(k_pklshl, I_PKLSHL)
KERNEL(k_snop, I_SNOP)
KERNEL(k_and, I_AND)
KERNEL(k_or, I_OR)
KERNEL(k_xor, I_XOR)
KERNEL(k_lshl, I_LSHL)
KERNEL(k_sub32, I_SUB32)
KERNEL(k_add16, I_ADD16)
KERNEL(k_mini16, I_MINI16)
KERNEL(k_maxu32, I_MAXU32)
KERNEL(k_minu32, I_MINU32)
KERNEL(k_lshlor, I_LSHLOR)
KERNEL(k_andor, I_ANDOR)
KERNEL(k_cmp, I_CMP)
KERNEL(k_cnd64, I_CND64)
KERNEL(k_addi16c, I_ADDI16C)
KERNEL(k_bfe, I_BFE)
KERNEL(k_add3, I_ADD3)

// one dependent chain per wave: what a lone wave pays per DEPENDENT instruction (issue + latency)
#define DEP8(INS) asm volatile(INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) : "+v"(x[0]) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b))
#define I_PKADD_D(k)  "v_pk_add_i16 %0, %0, %1 clamp\n\t"
#define I_ADD32_D(k)  "v_add_u32 %0, %0, %1\n\t"
#define I_PKADD_NOP_D(k)  "v_pk_add_i16 %0, %0, %1 clamp\n\ts_nop 0\n\t"
#define KERNEL_DEP(NAME, INS) \
__global__ __launch_bounds__(64) void NAME(uint32_t *out, uint32_t a, uint32_t b) { \
	uint32_t x[1] = { a + threadIdx.x }; \
	for (int it = 0; it < ITER; ++it) { \
		_Pragma("unroll") for (int r = 0; r < REP; ++r) DEP8(INS); \
	} \
	if (x[0] == 0x12345) out[threadIdx.x] = x[0]; \
}
KERNEL_DEP(k_dep_pkadd, I_PKADD_D)
KERNEL_DEP(k_dep_add32, I_ADD32_D)
KERNEL_DEP(k_dep_pkadd_nop, I_PKADD_NOP_D)

__global__ void k_clock(unsigned long long *out)
{
	const unsigned long long r0 = wall_clock64(), c0 = clock64();
	uint32_t x = threadIdx.x;
	for (int it = 0; it < 200000; ++it) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
	const unsigned long long r1 = wall_clock64(), c1 = clock64();
	if (threadIdx.x == 0) out[0] = r1 - r0, out[1] = c1 - c0, out[2] = x;
}

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t);
struct Case { const char *name; kern_t k; };

int main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int n_simd = p.multiProcessorCount * 4;
	uint32_t *out;
	hipMalloc(&out, 4096);
	unsigned long long *clk, hclk[3];
	hipMalloc(&clk, 64);
	k_clock<<<1, 64>>>(clk);
	hipMemcpy(hclk, clk, 24, hipMemcpyDeviceToHost);
	// wall_clock64 ticks at 100 MHz
	const double f_ghz = (double)hclk[1] / ((double)hclk[0] * 10.0);
	printf("# %s: %d CUs, clockRate %.0f MHz; lone-wave clock64 / wall_clock64 -> %.3f GHz (clock64 may count at a fixed rate)\n", p.name, p.multiProcessorCount, p.clockRate / 1000.0, f_ghz);
	const Case cases[] = {
		{ "v_fma_f32", k_fma32 }, { "v_add_u32", k_add32 }, { "v_max_i32", k_max32 }, { "v_mov_b32", k_mov }, { "v_pk_add_i16 clamp", k_pkadd_clamp }, { "v_pk_add_i16", k_pkadd },
		{ "v_pk_max_i16", k_pkmax }, { "v_pk_min_i16", k_pkmin }, { "v_pk_mad_i16", k_pkmad }, { "v_pk_lshlrev_b16", k_pklshl }, { "v_max_i16", k_maxi16 },
		{ "v_perm_b32", k_perm }, { "v_mov_b32_dpp row_shr:1", k_movdpp }, { "v_max_i32_dpp row_shr:1", k_maxdpp },
		{ "v_add_u32_sdwa", k_sdwa }, { "v_max3_i32", k_max3 }, { "v_med3_i32", k_med3 }, { "v_cndmask_b32 (vcc)", k_cndmask }, { "v_cndmask_b32_e64 (sgpr pair)", k_cnd64 }, { "s_nop 0", k_snop },
		{ "v_and_b32", k_and }, { "v_or_b32", k_or }, { "v_xor_b32", k_xor }, { "v_lshlrev_b32", k_lshl }, { "v_sub_u32", k_sub32 }, { "v_add_u16", k_add16 }, { "v_min_i16", k_mini16 },
		{ "v_max_u32", k_maxu32 }, { "v_min_u32", k_minu32 }, { "v_lshl_or_b32", k_lshlor }, { "v_and_or_b32", k_andor }, { "v_cmp_ne_u32 -> vcc", k_cmp }, { "v_add_i16 clamp (VOP3)", k_addi16c },
		{ "v_bfe_i32", k_bfe }, { "v_add3_u32", k_add3 },
		{ "DEPENDENT v_pk_add_i16 clamp", k_dep_pkadd }, { "DEPENDENT v_add_u32", k_dep_add32 }, { "DEPENDENT v_pk_add_i16 clamp + s_nop 0", k_dep_pkadd_nop },
	};
	printf("# cycles per wave64 instruction per SIMD at 2.4 GHz nominal (time * 2.4e9 / (instructions per wave * waves per SIMD)); 8 independent chains per wave unless DEPENDENT\n");
	printf("%-42s %8s %8s %8s %8s\n", "instruction", "W=1", "W=2", "W=4", "W=8");
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	for (const Case &c : cases) {
		printf("%-42s", c.name);
		const bool dep = strncmp(c.name, "DEPENDENT", 9) == 0, nop2 = strstr(c.name, "+ s_nop") != nullptr;
		for (int W = 1; W <= 8; W <<= 1) {
			const int grid = n_simd * W;                         // one-wave workgroups: the dispatcher spreads them over the SIMDs
			c.k<<<grid, 64>>>(out, 1, 2);                        // warm-up
			hipDeviceSynchronize();
			float best = 1e30f;
			for (int rep = 0; rep < 3; ++rep) {
				hipEventRecord(e0);
				c.k<<<grid, 64>>>(out, 1, 2);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				best = ms < best ? ms : best;
			}
			const double n_ins = (double)ITER * REP * 8 * (dep && nop2 ? 1 : 1);
			printf(" %8.2f", best * 1e-3 * 2.4e9 / (n_ins * W));
		}
		printf("\n");
	}
	return 0;
}
