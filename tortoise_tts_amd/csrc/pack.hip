// Weight packing, run once in ttk_*_create: f32 reference-layout tensors -> T-typed, padded, K-contiguous matrices
// (and the MFMA-fragment order of the decode path).
#include "ttk_common.h"
#include "ttk_kernels.h"

namespace ttk {

// dst[tap][n][k] (Npad x Kpad, zero padded) from
//   PK_NK    src[n][k]          nn.Linear [out,in], nn.Conv1d k=1 [out,in,1]
//   PK_KN    src[k][n]          HF Conv1D [in,out]   (HF:pytorch_utils.py:95-120)
//   PK_CONV3 src[n][k][3]       nn.Conv1d k=3 [out,in,3] -> 3 tap matrices, tap 0 multiplies row t-1
template <typename T>
__global__ void k_pack_nk(const float* src, int layout, int N, int K, int Npad, int Kpad, T* dst) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int ntap = layout == PK_CONV3 ? 3 : 1;
	const int64_t per = (int64_t)Npad * Kpad;
	if (idx >= ntap * per) return;
	const int tap = (int)(idx / per);
	const int64_t r = idx - tap * per;
	const int n = (int)(r / Kpad), k = (int)(r - (int64_t)n * Kpad);
	float v = 0.f;
	if (n < N && k < K) {
		if (layout == PK_NK) v = src[(int64_t)n * K + k];
		else if (layout == PK_KN) v = src[(int64_t)k * N + n];
		else v = src[((int64_t)n * K + k) * 3 + tap];
	}
	dst[idx] = cvt<T>(v);
}
void launch_pack_nk(int dt, const float* src, int layout, int N, int K, int Npad, int Kpad, void* dst, hipStream_t s) {
	const int64_t total = (int64_t)(layout == PK_CONV3 ? 3 : 1) * Npad * Kpad;
	const unsigned grid = (unsigned)((total + 255) / 256);
	if (dt == DT_BF16) hipLaunchKernelGGL((k_pack_nk<bf16>), dim3(grid), dim3(256), 0, s, src, layout, N, K, Npad, Kpad, (bf16*)dst);
	else hipLaunchKernelGGL((k_pack_nk<float>), dim3(grid), dim3(256), 0, s, src, layout, N, K, Npad, Kpad, (float*)dst);
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
	else hipLaunchKernelGGL((k_pack_frag<float>), dim3(grid), dim3(256), 0, s, (const float*)src, Npad, K, (float*)dst);
}

}  // namespace ttk
