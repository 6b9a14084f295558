// FETCH_SIZE / WRITE_SIZE calibration: kernels that move a KNOWN number of bytes with 2-, 4-, 8- and 16-byte accesses per lane
// (coalesced: consecutive lanes, consecutive addresses), each launched alone so that rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE rows can be
// divided by the bytes really moved.  The buffer (1 GiB) is larger than the Infinity Cache and every byte is touched once per launch.
// Build: hipcc --offload-arch=gfx950 -O3 profiles/micro/fetch_calib.hip -o /tmp/fetch_calib ; run under rocprofv3 --kernel-trace --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <typename V>
__global__ void read_kernel(const V *p, size_t n, uint32_t *sink)
{
	uint32_t acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const V v = p[i];
		const uint32_t *w = (const uint32_t*)&v;
		for (unsigned k = 0; k < (sizeof(V) + 3) / 4; ++k) acc ^= sizeof(V) >= 4 ? w[k] : (uint32_t)*(const uint16_t*)&v;
	}
	if (acc == 0x1234u) *sink = acc; // (practically) never true: keeps the loads
}
template <typename V>
__global__ void write_kernel(V *p, size_t n)
{
	V v;
	__builtin_memset(&v, 1, sizeof(V));
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
struct alignas(8) B8 { uint32_t a, b; };
struct alignas(16) B16 { uint32_t a, b, c, d; };

int main()
{
	const size_t bytes = (size_t)1 << 30;
	void *buf; uint32_t *sink;
	if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
	hipMemset(buf, 0, bytes);
	const int grid = 256 * 8, block = 256;
	for (int rep = 0; rep < 3; ++rep) {
		read_kernel<uint16_t><<<grid, block>>>((const uint16_t*)buf, bytes / 2, sink);
		read_kernel<uint32_t><<<grid, block>>>((const uint32_t*)buf, bytes / 4, sink);
		read_kernel<B8><<<grid, block>>>((const B8*)buf, bytes / 8, sink);
		read_kernel<B16><<<grid, block>>>((const B16*)buf, bytes / 16, sink);
		write_kernel<uint16_t><<<grid, block>>>((uint16_t*)buf, bytes / 2);
		write_kernel<uint32_t><<<grid, block>>>((uint32_t*)buf, bytes / 4);
		write_kernel<B8><<<grid, block>>>((B8*)buf, bytes / 8);
		write_kernel<B16><<<grid, block>>>((B16*)buf, bytes / 16);
	}
	hipDeviceSynchronize();
	printf("each kernel moves %zu bytes per launch\n", bytes);
	return 0;
}
