// What a uniform branch costs a wave (cycles), taken and not taken, alone and with co-resident waves: the packed band kernel
// spends one instruction in twelve on s_cbranch (profiles/r05: SQ_INSTS_BRANCH 7.8e8 of 9.6e9 per launch).
// Build: hipcc --offload-arch=gfx950 -O3 profiles/micro/branch_rates.hip -o profiles/micro/_branch_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))
#define REP64(X) REP4(REP16(X))

// WHICH 0: 8 VALU, no branch | 1: 8 VALU + a NOT-taken s_cbranch_scc1 | 2: 8 VALU + a TAKEN s_cbranch_scc1 over 8 more (skipped) VALU
//       3: 8 VALU + taken branch over 64 instructions (beyond the fetch window) | 4: v_cmp -> vcc -> s_cbranch_vccnz not taken
//       5: v_readlane -> s_cmp -> s_cbranch not taken | 6: s_and_saveexec + s_cbranch_execz (not taken) + s_or exec
template <int WHICH>
__global__ void k(int *out, long long *cyc, int seed)
{
	int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 ^ 5, a3 = a0 + 9, b = seed * 5 + 1;
	int sc = seed; // uniform, nonzero
	long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < 64; ++it) {
#define VALU8 "v_add_u32 %0, %0, %1\n v_max_i32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_xor_b32 %4, %4, %1\n v_add_u32 %0, %0, %1\n v_max_i32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_xor_b32 %4, %4, %1\n"
#define OPS : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sc) : "scc", "vcc", "s20", "s21", "s22", "s23"
		if (WHICH == 0) { REP64(asm volatile(VALU8 OPS);) }
		if (WHICH == 1) { REP64(asm volatile(VALU8 "s_cmp_eq_u32 %5, 0\n s_cbranch_scc1 1f\n 1:\n" OPS);) }
		if (WHICH == 2) { REP64(asm volatile(VALU8 "s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1f\n" VALU8 "1:\n" OPS);) }
		if (WHICH == 3) { REP64(asm volatile(VALU8 "s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1f\n" VALU8 VALU8 VALU8 VALU8 VALU8 VALU8 VALU8 VALU8 "1:\n" OPS);) }
		if (WHICH == 4) { REP64(asm volatile(VALU8 "v_cmp_eq_u32 vcc, 0x7fffffff, %0\n s_cbranch_vccnz 1f\n 1:\n" OPS);) }
		if (WHICH == 5) { REP64(asm volatile(VALU8 "v_readlane_b32 s20, %0, 3\n s_cmp_eq_u32 s20, 0x12345\n s_cbranch_scc1 1f\n 1:\n" OPS);) }
		if (WHICH == 6) { REP64(asm volatile(VALU8 "v_cmp_ne_u32 vcc, 0x7fffffff, %0\n s_and_saveexec_b64 s[22:23], vcc\n s_cbranch_execz 1f\n 1:\n s_or_b64 exec, exec, s[22:23]\n" OPS);) }
	}
	long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W>
double run(int waves_per_simd)
{
	int *out; long long *cyc;
	const int grid = 256, block = 256 * waves_per_simd;
	hipMalloc(&out, grid * block * 4); hipMalloc(&cyc, grid * 8);
	k<W><<<grid, block>>>(out, cyc, 1);
	hipDeviceSynchronize();
	k<W><<<grid, block>>>(out, cyc, 2);
	hipDeviceSynchronize();
	std::vector<long long> h(grid);
	hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
	double m = 0; for (auto v : h) m += v; m /= grid;
	hipFree(out); hipFree(cyc);
	return m / (64.0 * 64); // cycles per group (8 VALU + the branch sequence)
}

int main()
{
	const char *names[7] = {"8 VALU (base)", "+ s_cmp, s_cbranch_scc1 NOT taken", "+ s_cmp, s_cbranch_scc1 TAKEN over 8 VALU", "+ s_cmp, s_cbranch_scc1 TAKEN over 64 VALU",
	                        "+ v_cmp vcc, s_cbranch_vccnz not taken", "+ v_readlane, s_cmp, s_cbranch_scc1 not taken", "+ v_cmp, s_and_saveexec, s_cbranch_execz (not taken), s_or exec"};
	for (int w : {1, 2, 4}) {
		const double base = run<0>(w);
		const double r[7] = {base, run<1>(w), run<2>(w), run<3>(w), run<4>(w), run<5>(w), run<6>(w)};
		for (int i = 0; i < 7; ++i) printf("waves/SIMD %d: %-66s %7.1f cycles per group, %+6.1f over base\n", w, names[i], r[i], r[i] - base);
	}
	return 0;
}
