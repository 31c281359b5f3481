// Weight packing, run once in ttk_*_create: f32 reference-layout tensors -> T-typed, padded, K-contiguous matrices
// (and the MFMA-fragment order of the decode path).
#include <math.h>
#include <string.h>

#include "ttk_common.h"
#include "ttk_kernels.h"

namespace ttk {

// dst[tap][n][k] (Npad x Kpad, zero padded) from
//   PK_NK    src[n][k]          nn.Linear [out,in], nn.Conv1d k=1 [out,in,1]
//   PK_KN    src[k][n]          HF Conv1D [in,out]   (HF:pytorch_utils.py:95-120)
//   PK_CONV3 src[n][k][3]       nn.Conv1d k=3 [out,in,3] -> 3 tap matrices, tap 0 multiplies row t-1
//   PK_CONVK src[n][k][ntap]    nn.Conv1d of any odd kernel size
//   PK_CONVT src[k][n][ntap]    nn.ConvTranspose1d [in,out,taps]
template <typename T>
__global__ void k_pack_nk(const float* src, int layout, int N, int K, int Npad, int Kpad, int ntap, T* dst) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t per = (int64_t)Npad * Kpad;
	if (idx >= ntap * per) return;
	const int tap = (int)(idx / per);
	const int64_t r = idx - tap * per;
	const int n = (int)(r / Kpad), k = (int)(r - (int64_t)n * Kpad);
	float v = 0.f;
	if (n < N && k < K) {
		if (layout == PK_NK) v = src[(int64_t)n * K + k];
		else if (layout == PK_KN) v = src[(int64_t)k * N + n];
		else if (layout == PK_CONVT) v = src[((int64_t)k * N + n) * ntap + tap];       // ConvTranspose1d [in][out][taps]
		else v = src[((int64_t)n * K + k) * ntap + tap];                               // Conv1d [out][in][taps]
	}
	dst[idx] = cvt<T>(v);
}
void launch_pack_nk(int dt, const float* src, int layout, int N, int K, int Npad, int Kpad, void* dst, hipStream_t s, int ntap) {
	if (ntap <= 0) ntap = layout == PK_CONV3 ? 3 : 1;
	const int64_t total = (int64_t)ntap * Npad * Kpad;
	const unsigned grid = (unsigned)((total + 255) / 256);
	if (dt == DT_BF16) hipLaunchKernelGGL((k_pack_nk<bf16>), dim3(grid), dim3(256), 0, s, src, layout, N, K, Npad, Kpad, ntap, (bf16*)dst);
	else if (dt == DT_F16) hipLaunchKernelGGL((k_pack_nk<f16>), dim3(grid), dim3(256), 0, s, src, layout, N, K, Npad, Kpad, ntap, (f16*)dst);
	else hipLaunchKernelGGL((k_pack_nk<float>), dim3(grid), dim3(256), 0, s, src, layout, N, K, Npad, Kpad, ntap, (float*)dst);
}

// [Npad][K] -> Wp[n_tile][k_step][lane][8]: lane l holds W[16nt + (l&15)][32ks + 8(l>>4) + j]
template <typename T>
__global__ void k_pack_frag(const T* src, int Npad, int K, T* dst) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (int64_t)Npad * K) return;
	const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
	const int64_t t = idx >> 9;
	const int KS = K / 32;
	const int ks = (int)(t % KS), nt = (int)(t / KS);
	dst[idx] = src[(int64_t)(16 * nt + (lane & 15)) * K + 32 * ks + 8 * (lane >> 4) + j];
}
void launch_pack_frag(int dt, const void* src, int Npad, int K, void* dst, hipStream_t s) {
	const int64_t total = (int64_t)Npad * K;
	const unsigned grid = (unsigned)((total + 255) / 256);
	if (dt == DT_BF16) hipLaunchKernelGGL((k_pack_frag<bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)src, Npad, K, (bf16*)dst);
	else if (dt == DT_F16) hipLaunchKernelGGL((k_pack_frag<f16>), dim3(grid), dim3(256), 0, s, (const f16*)src, Npad, K, (f16*)dst);
	else hipLaunchKernelGGL((k_pack_frag<float>), dim3(grid), dim3(256), 0, s, (const float*)src, Npad, K, (float*)dst);
}

// ------------------------------------------------------------------------------------------------ LayerNorm folded into a decode GEMV
// LN(x) W + b  =  rstd * (x (gamma o W) - mean * colsum(gamma o W)) + (b + beta W):  the decode step's ln_1 + c_attn and ln_2 + c_fc launches
// multiply the UN-normalised row by W' = gamma o W and finish the LayerNorm in their epilogue from the row's (mean, rstd) -- see k_skinny's
// FOLD path.  w is the HF Conv1D weight [K][N] (f32, device).
__global__ void k_scale_kn(float* w, const float* gamma, int K, int N) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < (int64_t)K * N) w[i] *= gamma[i / N];
}
void launch_scale_kn(float* w, const float* gamma, int K, int N, hipStream_t s) {
	hipLaunchKernelGGL(k_scale_kn, dim3((unsigned)(((int64_t)K * N + 255) / 256)), dim3(256), 0, s, w, gamma, K, N);
}
// csum[n] = sum_k W'[n][k] over the ROUNDED (T-typed) values the kernel multiplies with, so that mean * csum cancels exactly what the
// MFMAs summed; one wave per row, f32 partial sums over 16 elements per lane, f64 across lanes
template <typename T>
__global__ void k_rowsum(const T* w, int64_t ld, int N, int K, float* out) {
	const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (n >= N) return;
	double acc = 0.0;
	for (int k = lane; k < K; k += 64) acc += (double)(float)w[(int64_t)n * ld + k];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
	if (lane == 0) out[n] = (float)acc;
}
void launch_rowsum(int dt, const void* w, int64_t ld, int N, int K, float* out, hipStream_t s) {
	if (dt == DT_BF16) hipLaunchKernelGGL((k_rowsum<bf16>), dim3((N + 3) / 4), dim3(256), 0, s, (const bf16*)w, ld, N, K, out);
	else if (dt == DT_F16) hipLaunchKernelGGL((k_rowsum<f16>), dim3((N + 3) / 4), dim3(256), 0, s, (const f16*)w, ld, N, K, out);
	else hipLaunchKernelGGL((k_rowsum<float>), dim3((N + 3) / 4), dim3(256), 0, s, (const float*)w, ld, N, K, out);
}
// bias'[n] = b[n] + sum_k beta[k] * W[k][n]   (f32 weights as given, f64 accumulation)
__global__ void k_bias_fold(const float* w, const float* beta, const float* b, int K, int N, float* out) {
	const int n = blockIdx.x * blockDim.x + threadIdx.x;
	if (n >= N) return;
	double acc = b ? (double)b[n] : 0.0;
	for (int k = 0; k < K; ++k) acc += (double)beta[k] * (double)w[(int64_t)k * N + n];
	out[n] = (float)acc;
}
void launch_bias_fold(const float* w, const float* beta, const float* b, int K, int N, float* out, hipStream_t s) {
	hipLaunchKernelGGL(k_bias_fold, dim3((N + 255) / 256), dim3(256), 0, s, w, beta, b, K, N, out);
}

// ------------------------------------------------------------------------------------------------ fp8-e4m3 weights
// Quantisation is done BY the hardware conversion instructions (v_cvt_pk_fp8_f32 / v_cvt_pk_f32_fp8: OCP e4m3, round to nearest
// even, finite range +-448), so what the decode kernels decode is by construction what was encoded here; tests pin the format against
// torch.float8_e4m3fn.  The per-tensor scale is a power of two >= absmax / 448: dividing and multiplying by it is exact, and every
// dequantised weight (3 mantissa bits) is exactly representable in bf16, so "fp8 weights" is bit-for-bit "bf16 kernels on rounded weights".
__global__ void k_absmax(const float* x, int64_t n, unsigned* out) {
	float m = 0.f;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
	m = wave_max(m);
	if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bit patterns
}
int device_absmax(const float* x, int64_t n, float* out_host) {
	unsigned* d = nullptr;
	if (hipMalloc((void**)&d, 4) != hipSuccess) return -1;
	(void)hipMemset(d, 0, 4);
	hipLaunchKernelGGL(k_absmax, dim3(512), dim3(256), 0, 0, x, n, d);
	unsigned bits = 0;
	const hipError_t e = hipMemcpy(&bits, d, 4, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return -1;
	memcpy(out_host, &bits, 4);
	return 0;
}
float fp8_scale_for(float absmax) {
	if (!(absmax > 0.f) || !isfinite(absmax)) return 1.f;
	int e;
	const float fr = frexpf(absmax / 448.0f, &e);         // absmax / 448 = fr * 2^e, fr in [0.5, 1)
	return ldexpf(1.0f, fr == 0.5f ? e - 1 : e);          // smallest power of two >= absmax / 448
}
__device__ __forceinline__ float fp8_round(float v) {
	typedef float f2 __attribute__((ext_vector_type(2)));
	const int q = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
	const f2 r = __builtin_amdgcn_cvt_pk_f32_fp8(q, false);
	return r[0];
}
__global__ void k_fp8_roundtrip(float* x, int64_t n, float scale, float inv) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) x[i] = fp8_round(x[i] * inv) * scale;
}
void launch_fp8_roundtrip(float* x, int64_t n, float scale, hipStream_t s) {
	hipLaunchKernelGGL(k_fp8_roundtrip, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, scale, 1.0f / scale);
}
// the dense fp8 GEMM's weight operand: same index mapping as k_pack_nk, one fp8-e4m3 byte per element (value / scale, exact: the source was
// rounded to scale * fp8 grid by k_fp8_roundtrip)
__global__ void k_pack_nk_f8(const float* src, int layout, int N, int K, int Npad, int Kpad, int ntap, float inv, unsigned char* dst) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t per = (int64_t)Npad * Kpad;
	if (idx >= ntap * per) return;
	const int tap = (int)(idx / per);
	const int64_t r = idx - tap * per;
	const int n = (int)(r / Kpad), k = (int)(r - (int64_t)n * Kpad);
	float v = 0.f;
	if (n < N && k < K) {
		if (layout == PK_NK) v = src[(int64_t)n * K + k];
		else if (layout == PK_KN) v = src[(int64_t)k * N + n];
		else if (layout == PK_CONVT) v = src[((int64_t)k * N + n) * ntap + tap];
		else v = src[((int64_t)n * K + k) * ntap + tap];
	}
	dst[idx] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v * inv, 0.f, 0, false) & 0xff);
}
void launch_pack_nk_f8(const float* src, int layout, int N, int K, int Npad, int Kpad, float scale, void* dst, hipStream_t s, int ntap) {
	if (ntap <= 0) ntap = layout == PK_CONV3 ? 3 : 1;
	const int64_t total = (int64_t)ntap * Npad * Kpad;
	hipLaunchKernelGGL(k_pack_nk_f8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, layout, N, K, Npad, Kpad, ntap, 1.0f / scale, (unsigned char*)dst);
}
// [Npad][K] bf16 (values = fp8 grid * scale) -> Wp8[n_tile][k_step][lane][8 bytes], same element order as k_pack_frag
__global__ void k_pack_frag_fp8(const bf16* src, int Npad, int K, float inv, unsigned char* dst) {
	const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;      // two elements (one 16-bit half) per thread
	if (idx >= (int64_t)Npad * K) return;
	const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
	const int64_t t = idx >> 9;
	const int KS = K / 32;
	const int ks = (int)(t % KS), nt = (int)(t / KS);
	const bf16* p = src + (int64_t)(16 * nt + (lane & 15)) * K + 32 * ks + 8 * (lane >> 4) + j;
	const int q = __builtin_amdgcn_cvt_pk_fp8_f32((float)p[0] * inv, (float)p[1] * inv, 0, false);
	*(unsigned short*)(dst + idx) = (unsigned short)(q & 0xffff);
}
void launch_pack_frag_fp8(const void* src, int Npad, int K, float scale, void* dst, hipStream_t s) {
	const int64_t total = (int64_t)Npad * K / 2;
	hipLaunchKernelGGL(k_pack_frag_fp8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16*)src, Npad, K, 1.0f / scale, (unsigned char*)dst);
}

}  // namespace ttk
