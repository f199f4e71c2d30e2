// Does a kernel launched with hipExtAnyOrderLaunch overlap the kernel in front of it on the same stream?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long *out)
{
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
	}
	if (out && threadIdx.x == 0)
		out[blockIdx.x] = t0;
}
int main()
{
	hipStream_t s;
	hipStreamCreate(&s);
	unsigned long long *d;
	hipMalloc(&d, 4096);
	const unsigned long long ticks = 3000;   // 30 us at 100 MHz
	for (int mode = 0; mode < 3; mode++) {
		for (int rep = 0; rep < 3; rep++) {
			hipStreamSynchronize(s);
			auto t0 = std::chrono::steady_clock::now();
			for (int k = 0; k < 8; k++) {
				if (mode == 0 || (mode == 2 && (k & 1) == 0))
					hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, d);
				else
					hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d);
			}
			hipStreamSynchronize(s);
			const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
			printf("mode %d (%s): 8 kernels of 30 us took %.1f us\n", mode,
			    mode == 0 ? "in order" : mode == 1 ? "all any-order" : "every second any-order", us);
		}
	}
	return 0;
}
