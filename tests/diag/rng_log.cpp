// Diagnostic: which device log reproduces torch's exponential_ (ATen at::log -> __logf as compiled into the torch wheel)?
//   hipcc --offload-arch=gfx950 -O3 tests/diag/rng_log.cpp -o tests/diag/rng_log.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
extern "C" __device__ float __ocml_native_log_f32(float);
extern "C" __device__ float __ocml_log_f32(float);
__global__ void k(const float* u, float* out, int n) {
	int i = threadIdx.x;
	if (i >= n) return;
	out[0 * n + i] = -logf(u[i]);
	out[1 * n + i] = -__logf(u[i]);
	out[2 * n + i] = -(__builtin_amdgcn_logf(u[i]) * 0x1.62e430p-1f);
	out[3 * n + i] = -__ocml_native_log_f32(u[i]);
	out[4 * n + i] = -(__builtin_log2f(u[i]) * 0x1.62e430p-1f);
	out[5 * n + i] = -__ocml_log_f32(u[i]);
	out[6 * n + i] = -1.0f / 1.0f * (__builtin_amdgcn_logf(u[i]) * 0x1.62e430p-1f);
}
int main() {
	const int n = 4;
	float hu[n] = {0.39904648065567017f, 0.5166791677474976f, 0.024930385872721672f, 0.940079391002655f};
	float *u, *o; hipMalloc(&u, sizeof hu); hipMalloc(&o, 7 * sizeof hu);
	hipMemcpy(u, hu, sizeof hu, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, u, o, n);
	float ho[7 * n]; hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
	const char* names[7] = {"logf", "__logf", "amdgcn_logf*ln2", "ocml_native_log", "builtin_log2f*ln2", "ocml_log", "-1/1*(amdgcn_logf*ln2)"};
	printf("torch: 0.9186774492263794 0.6603332161903381 3.6916680335998535 0.06179095059633255\n");
	for (int v = 0; v < 7; ++v) { printf("%-24s", names[v]); for (int i = 0; i < n; ++i) printf(" %.16g", ho[v * n + i]); printf("\n"); }
	return 0;
}
