// How long hipMalloc / hipFree take by size: what the first CIGAR-mode call of an engine pays for its traceback arena (mwf_plan.cpp tb_budget_bytes).
// Build: hipcc --offload-arch=gfx950 -O2 profiles/micro/malloc_time.hip -o gpurun_out/malloc_time ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main()
{
	hipFree(0);
	for (int rep = 0; rep < 2; ++rep)
		for (long gb : {1L, 2L, 4L, 8L, 16L, 32L, 64L}) {
			void *p = nullptr;
			auto t0 = std::chrono::steady_clock::now();
			hipError_t e = hipMalloc(&p, (size_t)gb << 30);
			auto t1 = std::chrono::steady_clock::now();
			if (e != hipSuccess) { printf("%ld GB: %s\n", gb, hipGetErrorString(e)); continue; }
			hipMemset(p, 0, 4096);
			hipDeviceSynchronize();
			auto t2 = std::chrono::steady_clock::now();
			hipFree(p);
			auto t3 = std::chrono::steady_clock::now();
			auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
			printf("%ld GB: hipMalloc %.1f ms, first memset %.1f ms, hipFree %.1f ms\n", gb, ms(t0, t1), ms(t1, t2), ms(t2, t3));
		}
	return 0;
}
