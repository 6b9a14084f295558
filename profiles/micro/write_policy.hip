// Does a store that hits a line already dirty in L2 reach HBM again?  A 1 MiB buffer (fits any L2 slice many times over) is rewritten REPS times by
// the same threads, with a device-wide pause between rounds (so that the stores cannot merge in the L1 write path).  rocprofv3 --pmc WRITE_SIZE:
// ~1 MiB = write-back (rewrites stay in L2), ~REPS MiB = every round of stores reaches memory.  Round 6: asked because the whole-device kernel's
// WRITE_SIZE (14.3 GB on the 150 kb pair) did not move when its hand-off boxes were laid out per penalty.
// Build: hipcc --offload-arch=gfx950 -O3 profiles/micro/write_policy.hip -o profiles/micro/_write_policy ; run under rocprofv3 --kernel-trace --pmc WRITE_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void rewrite(int *buf, int n, int reps, int salt)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	for (int r = 0; r < reps; ++r) {
		if (i < n) buf[i] = r * 7 + i + salt;
		__builtin_amdgcn_s_sleep(127);
		__builtin_amdgcn_s_sleep(127);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
}
__global__ void rewrite_sc1(int *buf, int n, int reps, int salt)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	for (int r = 0; r < reps; ++r) {
		if (i < n) __hip_atomic_store(buf + i, r * 7 + i + salt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__builtin_amdgcn_s_sleep(127);
		__builtin_amdgcn_s_sleep(127);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
}
int main()
{
	const int n = 1 << 18, reps = 64; // 1 MiB of ints
	int *buf;
	hipMalloc(&buf, n * 4);
	hipMemset(buf, 0, n * 4);
	for (int k = 0; k < 3; ++k) rewrite<<<n / 256, 256>>>(buf, n, reps, k);
	hipDeviceSynchronize();
	for (int k = 0; k < 3; ++k) rewrite_sc1<<<n / 256, 256>>>(buf, n, reps, k);
	hipDeviceSynchronize();
	printf("wrote %d MiB x %d rounds per launch, 3 launches per kernel\n", n * 4 >> 20, reps);
	return 0;
}
