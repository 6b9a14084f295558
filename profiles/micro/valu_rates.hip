// Issue cost (cycles per wave64 instruction on one SIMD) of the integer VALU forms the band kernel is made of.
// One workgroup of 256 threads per CU (one wave per SIMD), N dependent-free instructions per loop trip, s_memtime around.
// Build: hipcc --offload-arch=gfx950 -O3 profiles/micro/valu_rates.hip -o /tmp/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int WHICH>
__global__ void k(int *out, long long *cyc, int seed)
{
	int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 ^ 5, a3 = a0 + 9, a4 = a0 * 7, a5 = a0 - 3, a6 = a0 ^ 77, a7 = a0 + 100;
	int b = seed * 5 + 1, c = seed + 3;
	unsigned long long sm = 0;
	long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < 64; ++it) {
		if (WHICH == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
		if (WHICH == 1) { REP64(asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %2, %2, %1\n v_max_i32 %3, %3, %1\n v_max_i32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
		if (WHICH == 2) { REP64(asm volatile("v_max3_i32 %0, %0, %1, %5\n v_max3_i32 %2, %2, %1, %5\n v_max3_i32 %3, %3, %1, %5\n v_max3_i32 %4, %4, %1, %5" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
		if (WHICH == 3) { REP64(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %1, vcc\n v_cmp_gt_u32 vcc, %3, %1\n v_cndmask_b32 %4, %4, %1, vcc" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
		if (WHICH == 4) { REP64(asm volatile("v_cmp_gt_u32 s[20:21], %0, %1\n v_cmp_gt_u32 s[22:23], %2, %1\n v_cmp_gt_u32 s[24:25], %3, %1\n v_cmp_gt_u32 s[26:27], %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) :: "s20","s21","s22","s23","s24","s25","s26","s27");) }
		if (WHICH == 5) { REP64(asm volatile("v_max_i32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_max_i32_sdwa %2, %2, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_max_i32_sdwa %3, %3, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_max_i32_sdwa %4, %4, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
		if (WHICH == 6) { REP64(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
		if (WHICH == 7) { REP64(asm volatile("v_alignbyte_b32 %0, %0, %1, %2\n v_alignbyte_b32 %2, %2, %1, %3\n v_alignbyte_b32 %3, %3, %1, %4\n v_alignbyte_b32 %4, %4, %1, %0" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
		if (WHICH == 8) { REP64(asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %2, %2, %1\n v_pk_add_i16 %3, %3, %1\n v_pk_add_i16 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
		if (WHICH == 9) { REP64(asm volatile("v_ffbl_b32 %0, %1\n v_lshrrev_b32 %2, 3, %2\n v_min3_i32 %3, %3, %1, %0\n v_xor_b32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
		if (WHICH == 10) { REP64(asm volatile("v_cvt_pk_i16_i32 %0, %0, %1\n v_cvt_pk_i16_i32 %2, %2, %1\n v_cvt_pk_i16_i32 %3, %3, %1\n v_cvt_pk_i16_i32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
		if (WHICH == 11) { REP64(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %2, 5\n v_readlane_b32 s22, %3, 7\n v_readlane_b32 s23, %4, 9" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) :: "s20","s21","s22","s23");) }
		if (WHICH == 12) { REP64(asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[20:21], s[24:25]\n s_and_b64 s[24:25], s[20:21], s[22:23]\n s_add_i32 s26, s26, 1" ::: "s20","s21","s22","s23","s24","s25","s26", "scc");) }
		if (WHICH == 13) { REP64(asm volatile("v_min_u32 %0, %0, %1\n v_sub_u32 %2, %2, %1\n v_and_b32 %3, %3, %1\n v_add3_u32 %4, %4, %1, 3" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
		if (WHICH == 14) { REP64(asm volatile("v_pk_max_i16 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_max_i32 %3, %3, %1\n s_add_i32 s26, s26, 1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) :: "s26", "scc");) }
	}
	long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b + (int)sm;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W>
void run(const char *name, int waves_per_simd)
{
	int *out; long long *cyc;
	const int grid = waves_per_simd > 4 ? 512 : 256, block = waves_per_simd > 4 ? 1024 : 256 * waves_per_simd; // 8 waves per SIMD: two 1024-thread workgroups per CU
	hipMalloc(&out, grid * block * 4); hipMalloc(&cyc, grid * 8);
	k<W><<<grid, block>>>(out, cyc, 1);
	hipDeviceSynchronize();
	k<W><<<grid, block>>>(out, cyc, 2);
	hipDeviceSynchronize();
	std::vector<long long> h(grid);
	hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
	double m = 0; for (auto v : h) m += v; m /= grid;
	const double n_inst = 64.0 * 64 * 4; // per wave
	printf("%-44s waves/SIMD %d: %.2f cycles per instruction per wave, %.2f per SIMD issue slot\n", name, waves_per_simd, m / n_inst, m / n_inst / waves_per_simd);
	hipFree(out); hipFree(cyc);
}

int main()
{
	for (int w : {1, 2, 4, 8}) {
		run<0>("v_add_u32", w); run<1>("v_max_i32", w); run<2>("v_max3_i32", w); run<3>("v_cmp(vcc)+v_cndmask pairs", w);
		run<4>("v_cmp_gt_u32 -> sgpr pair", w); run<5>("v_max_i32_sdwa (sext word)", w); run<6>("v_mov_b32_dpp wave_shr:1", w);
		run<7>("v_alignbyte_b32", w); run<8>("v_pk_max_i16 / v_pk_add_i16", w); run<9>("ffbl/lshr/min3/xor mix", w);
		run<10>("v_cvt_pk_i16_i32", w); run<11>("v_readlane_b32", w); run<12>("SALU and/or b64 + add", w); run<13>("min_u32/sub/and/add3 mix", w);
		run<14>("pk_max + add + max + s_add interleaved", w);
	}
	return 0;
}
