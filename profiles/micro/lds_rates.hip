// Cost of the LDS / global access forms the balanced band kernel is made of, per wave64 instruction, at 1 / 2 / 4 waves per SIMD:
// consecutive-lane 2-byte reads and writes (ds_read_i16 / ds_write_b16), 4-byte ones, scattered ds_read2_b32, 2-byte global loads.
// Build: hipcc --offload-arch=gfx950 -O3 profiles/micro/lds_rates.hip -o /tmp/lds_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

extern __shared__ unsigned char lds[];
typedef __attribute__((address_space(3))) short lds_s16;
typedef __attribute__((address_space(3))) int lds_s32;

template <int WHICH>
__global__ void k(int *out, long long *cyc, const short *g, int seed)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int acc = seed;
	const unsigned base = wave * 4096;
	for (int j = threadIdx.x; j < 16384; j += blockDim.x) ((int*)lds)[j] = j * 7 + seed;
	__syncthreads();
	long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < 256; ++it) {
		if (WHICH == 0) { // 8 x ds_read_i16, consecutive lanes
#pragma unroll
			for (int u = 0; u < 8; ++u) acc += *(lds_s16*)(uintptr_t)(base + 2 * lane + 128 * u + ((acc & 0) | (it & 7) * 2));
		}
		if (WHICH == 1) { // 8 x ds_write_b16
#pragma unroll
			for (int u = 0; u < 8; ++u) *(lds_s16*)(uintptr_t)(base + 2 * lane + 128 * u + (it & 7) * 2) = (short)(acc + u);
			acc += it;
		}
		if (WHICH == 2) { // 8 x ds_read_b32 consecutive
#pragma unroll
			for (int u = 0; u < 8; ++u) acc += *(lds_s32*)(uintptr_t)(base + 4 * lane + 256 * u + (it & 7) * 4);
		}
		if (WHICH == 3) { // 8 x ds_write_b32
#pragma unroll
			for (int u = 0; u < 8; ++u) *(lds_s32*)(uintptr_t)(base + 4 * lane + 256 * u + (it & 7) * 4) = acc + u;
			acc += it;
		}
		if (WHICH == 4) { // 8 x scattered ds_read2_b32 (pseudo-random dword index, dependent chain broken per group of 8)
			unsigned h = (unsigned)(lane * 2654435761u + it * 40503u);
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const unsigned a = ((h >> (u * 3)) & 0x3ff) * 4;
				const lds_s32 *p = (const lds_s32*)(uintptr_t)(a);
				acc += p[0] ^ p[1];
			}
		}
		if (WHICH == 5) { // 8 x global_load_sshort, consecutive lanes, L2-resident rows
#pragma unroll
			for (int u = 0; u < 8; ++u) acc += g[(blockIdx.x * 16 + wave) * 4096 + lane + 64 * u + (it & 7)];
		}
		if (WHICH == 6) { // dependent: read i16 -> a few VALU -> scattered read2 -> VALU -> write b16 (the chunk's LDS round trips)
			int v = *(lds_s16*)(uintptr_t)(base + 2 * lane + (it & 7) * 2);
			v = (v * 3 + acc) & 0x3ff;
			const lds_s32 *p = (const lds_s32*)(uintptr_t)(v * 4);
			acc += p[0] ^ p[1];
			*(lds_s16*)(uintptr_t)(base + 2 * lane + 1024 + (it & 7) * 2) = (short)acc;
		}
	}
	long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W>
void run(const char *name, int waves_per_simd, double n_inst, const short *g)
{
	int *out; long long *cyc;
	const int grid = 256, block = 256 * waves_per_simd;
	hipMalloc(&out, grid * block * 4); hipMalloc(&cyc, grid * 8);
	k<W><<<grid, block, 65536>>>(out, cyc, g, 1);
	hipDeviceSynchronize();
	k<W><<<grid, block, 65536>>>(out, cyc, g, 2);
	hipDeviceSynchronize();
	std::vector<long long> h(grid);
	hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
	double m = 0; for (auto v : h) m += v; m /= grid;
	printf("%-58s waves/SIMD %d: %.1f cycles per iteration, %.1f per instruction of interest\n", name, waves_per_simd, m / 256, m / 256 / n_inst);
	hipFree(out); hipFree(cyc);
}

int main()
{
	short *g; hipMalloc(&g, (size_t)256 * 16 * 4096 * 2 + 4096); hipMemset(g, 1, (size_t)256 * 16 * 4096 * 2 + 4096);
	for (int w : {1, 2, 4}) {
		run<0>("8 x ds_read_i16 (consecutive lanes)", w, 8, g); run<1>("8 x ds_write_b16", w, 8, g); run<2>("8 x ds_read_b32", w, 8, g);
		run<3>("8 x ds_write_b32", w, 8, g); run<4>("8 x scattered ds_read2_b32", w, 8, g); run<5>("8 x global_load_sshort (L2)", w, 8, g);
		run<6>("dependent i16 read -> scattered read2 -> b16 write", w, 3, g);
	}
	return 0;
}
