// What a read-only stream reaches on this part: the ceiling the bulk kernel (k_sieve) is measured against besides
// the 8 TB/s of the spec.  2 GiB of device memory read once per launch, 16 bytes per lane and load, the loads'
// values folded into one word per lane so that nothing is optimised away.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/readbw tools/micro/readbw.hip && tools/micro/readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LOADS, bool NT>
__global__ void k_read(const v4u *p, size_t n16, unsigned *out)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned acc = 0;
	for (; i + (LOADS - 1) * stride < n16; i += LOADS * stride) {
		v4u v[LOADS];
#pragma unroll
		for (int k = 0; k < LOADS; k++)
			v[k] = NT ? __builtin_nontemporal_load(p + i + k * stride) : p[i + k * stride];
#pragma unroll
		for (int k = 0; k < LOADS; k++)
			acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
	}
	if (acc == 0x12345678u)
		out[0] = acc;
}

// the bulk kernel's shape: persistent workgroups, a wave owns 8 KiB tiles, eight loads of 1 KiB in flight
template <bool NT>
__global__ void k_tiles(const v4u *p, size_t n16, unsigned *out)
{
	const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
	const size_t ntiles = n16 / 512;
	unsigned acc = 0;
	for (size_t t = wave; t < ntiles; t += nwaves) {
		const v4u *q = p + t * 512 + lane;
		v4u v[8];
#pragma unroll
		for (int k = 0; k < 8; k++)
			v[k] = NT ? __builtin_nontemporal_load(q + k * 64) : q[k * 64];
#pragma unroll
		for (int k = 0; k < 8; k++)
			acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
	}
	if (acc == 0x12345678u)
		out[0] = acc;
}

template <typename F>
static double time_ms(F launch, int reps)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	launch();
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a, 0));
	for (int i = 0; i < reps; i++)
		launch();
	CK(hipEventRecord(b, 0));
	CK(hipEventSynchronize(b));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

int main()
{
	const size_t bytes = (size_t)2 << 30, n16 = bytes / 16;
	v4u *p;
	unsigned *out;
	CK(hipMalloc((void **)&p, bytes));
	CK(hipMalloc((void **)&out, 64));
	CK(hipMemset(p, 1, bytes));
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("%s, %d CUs; 2 GiB read once per launch\n", prop.name, cus);
	auto report = [&](const char *name, double ms) { printf("%-64s %8.1f GB/s\n", name, bytes / ms / 1e6); };
	for (int wg : { 1, 2, 4, 8 }) {
		char nm[96];
		snprintf(nm, sizeof nm, "grid-stride, 4 loads in flight, %d x 256 threads per CU", wg);
		report(nm, time_ms([&]() { k_read<4, false><<<dim3(cus * wg), dim3(256)>>>(p, n16, out); }, 10));
		snprintf(nm, sizeof nm, "grid-stride, 8 loads in flight, %d x 256 threads per CU", wg);
		report(nm, time_ms([&]() { k_read<8, false><<<dim3(cus * wg), dim3(256)>>>(p, n16, out); }, 10));
		snprintf(nm, sizeof nm, "grid-stride, 8 nontemporal loads in flight, %d x 256 per CU", wg);
		report(nm, time_ms([&]() { k_read<8, true><<<dim3(cus * wg), dim3(256)>>>(p, n16, out); }, 10));
	}
	report("tiles of 8 KiB per wave, 8 waves per CU (k_sieve's shape)", time_ms([&]() { k_tiles<false><<<dim3(cus), dim3(512)>>>(p, n16, out); }, 10));
	report("tiles of 8 KiB per wave, 16 waves per CU", time_ms([&]() { k_tiles<false><<<dim3(cus), dim3(1024)>>>(p, n16, out); }, 10));
	report("tiles of 8 KiB per wave, 2 x 8 waves per CU", time_ms([&]() { k_tiles<false><<<dim3(cus * 2), dim3(512)>>>(p, n16, out); }, 10));
	report("tiles, nontemporal, 8 waves per CU", time_ms([&]() { k_tiles<true><<<dim3(cus), dim3(512)>>>(p, n16, out); }, 10));
	report("tiles, nontemporal, 16 waves per CU", time_ms([&]() { k_tiles<true><<<dim3(cus), dim3(1024)>>>(p, n16, out); }, 10));
	return 0;
}
