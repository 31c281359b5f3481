// ttk_mel: the mel front-ends of the conditioning path behind the C ABI of include/ttk.h.
//   TorchMelSpectrogram  /root/reference/tortoise_tts/models/arch_utils.py:361-395  (power spectrogram, log, per-band divisor)
//   TacotronSTFT         /root/reference/tortoise_tts/models/arch_utils.py:662-700 over STFT :560-623 (clip, magnitude, log)
// Both: reflect-padded frames [b*F][n_fft] x windowed DFT matrix [2*(n_fft/2+1)][n_fft]^T -> |.|^p -> x mel matrix -> log(clamp 1e-5).
// The two GEMMs run in the exact-f32 mode of the dense kernel (a spectrogram spans > 100 dB; bf16 operands would bury the low bands);
// the matrices come from the host (tortoise_tts_amd/mel.py), so any window / mel definition maps onto the same three launches.
#include <stdlib.h>

#include "ttk_common.h"
#include "ttk_host.h"

using namespace ttk;

namespace {

// frame f of clip b: x[b][reflect(f * hop + j - n_fft / 2)], optionally clipped to [-1, 1]  (F.pad(mode='reflect'): no edge repeat)
__global__ __launch_bounds__(256) void k_mel_frames(const float* __restrict__ x, int n, int F, int n_fft, int hop, int clip, int64_t total,
													float* __restrict__ out) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= total) return;
	const int j = (int)(i % n_fft);
	const int64_t row = i / n_fft;
	const int f = (int)(row % F);
	const int64_t b = row / F;
	int idx = f * hop + j - n_fft / 2;
	if (idx < 0) idx = -idx;
	if (idx >= n) idx = 2 * (n - 1) - idx;
	float v = x[b * n + idx];
	if (clip) v = fminf(fmaxf(v, -1.f), 1.f);
	out[i] = v;
}

// spec f32 [rows][2 * nb] (Re | Im) -> |.|^power f32 [rows][Kpad], zero beyond nb
__global__ __launch_bounds__(256) void k_mel_mag(const float* __restrict__ spec, int nb, int Kpad, int power, int64_t total, float* __restrict__ out) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= total) return;
	const int k = (int)(i % Kpad);
	const int64_t row = i / Kpad;
	float v = 0.f;
	if (k < nb) {
		const float re = spec[row * 2 * nb + k], im = spec[row * 2 * nb + nb + k];
		v = re * re + im * im;
		if (power == 1) v = sqrtf(v);
	}
	out[i] = v;
}

// m f32 [b * F][n_mels] -> out [b][n_mels][F] = log(max(m, 1e-5)) / norms
__global__ __launch_bounds__(256) void k_mel_log(const float* __restrict__ m, int F, int n_mels, const float* __restrict__ norms, int64_t total,
												 float* __restrict__ out) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= total) return;
	const int f = (int)(i % F);
	const int64_t bm = i / F;
	const int band = (int)(bm % n_mels);
	const int64_t b = bm / n_mels;
	float v = (float)log((double)fmaxf(m[(b * F + f) * n_mels + band], 1e-5f));   // correctly rounded: the floor is exactly TACOTRON_MEL_MIN (arch_utils.py:533)
	if (norms) v = v / norms[band];
	out[i] = v;
}

// polyphase FIR resampling as torchaudio lays it out: output sample o = frame * gnew + phase reads K = 2 * width + gorig input samples
// starting at frame * gorig - width (zeros outside the clip) against row `phase` of the kernel table
__global__ __launch_bounds__(256) void k_resample_fir(const float* __restrict__ x, int n, const float* __restrict__ kernels, int gorig, int gnew,
													  int width, int K, int n_out, int64_t total, float* __restrict__ out) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= total) return;
	const int o = (int)(i % n_out);
	const int64_t b = i / n_out;
	const int phase = o % gnew, frame = o / gnew;
	const int base = frame * gorig - width;
	const float* kr = kernels + (int64_t)phase * K;
	const float* xr = x + b * n;
	float acc = 0.f;
	for (int k = 0; k < K; ++k) {
		const int j = base + k;
		if (j >= 0 && j < n) acc += kr[k] * xr[j];
	}
	out[i] = acc;
}

}  // namespace

struct ttk_mel {
	ttk_mel_config cfg;
	int nb;
	Arena arena;
	Mat basis, melb;
	float* norms = nullptr;
	WsBuf ws;
};

extern "C" {

int ttk_mel_create(ttk_mel** out, const ttk_mel_config* cfg, const ttk_weight_view* w, int n_w) {
	TTK_REQUIRE(out && cfg && w, TTK_E_ARG, "ttk_mel_create: null argument");
	TTK_REQUIRE(cfg->n_fft >= 64 && cfg->n_fft % 64 == 0 && cfg->hop >= 1 && cfg->n_mels >= 1, TTK_E_ARG,
				"ttk_mel_create: bad sizes (n_fft %d must be a multiple of 64, hop %d, n_mels %d)", cfg->n_fft, cfg->hop, cfg->n_mels);
	TTK_REQUIRE(cfg->power == 1 || cfg->power == 2, TTK_E_ARG, "ttk_mel_create: power must be 1 (magnitude) or 2, got %d", cfg->power);
	ttk_mel* h = new ttk_mel();
	h->cfg = *cfg;
	h->nb = cfg->n_fft / 2 + 1;
	WeightMap wm(w, n_w);
	int rc = TTK_OK;
	auto fail = [&](int code) { h->arena.release(); delete h; return code; };
#define M_TRY(expr) do { rc = (expr); if (rc != TTK_OK) return fail(rc); } while (0)
	M_TRY(upload_mat(h->arena, wm, DT_F32, "basis", "", PK_NK, 2 * h->nb, cfg->n_fft, false, &h->basis));
	M_TRY(upload_mat(h->arena, wm, DT_F32, "mel_basis", "", PK_NK, cfg->n_mels, h->nb, false, &h->melb));
	if (cfg->has_norms) M_TRY(upload_f32(h->arena, wm, "mel_norms", cfg->n_mels, &h->norms));
#undef M_TRY
	hipError_t e = hipDeviceSynchronize();
	if (e != hipSuccess) { set_error("ttk_mel_create: %s", hipGetErrorString(e)); return fail(TTK_E_HIP); }
	*out = h;
	return TTK_OK;
}

int ttk_mel_destroy(ttk_mel* h) {
	if (!h) return TTK_OK;
	(void)hipDeviceSynchronize();
	h->ws.release();
	h->arena.release();
	delete h;
	return TTK_OK;
}

int ttk_mel_forward(ttk_mel* h, const float* wav, int b, int n, float* mel, void* stream) {
	TTK_REQUIRE(h && wav && mel, TTK_E_ARG, "ttk_mel_forward: null argument");
	const ttk_mel_config& c = h->cfg;
	TTK_REQUIRE(b >= 1 && n > c.n_fft / 2, TTK_E_ARG, "ttk_mel_forward: a clip needs more than n_fft / 2 = %d samples (reflect padding), got b=%d n=%d",
				c.n_fft / 2, b, n);
	const int F = n / c.hop + 1;
	const int64_t rows = (int64_t)b * F;
	TTK_REQUIRE(rows * c.n_fft < ((int64_t)1 << 31), TTK_E_ARG, "ttk_mel_forward: %lld frames exceed one launch (split the batch)", (long long)rows);
	hipStream_t s = (hipStream_t)stream;
	const size_t fr = (size_t)rows * c.n_fft * 4, sp = (size_t)rows * 2 * h->nb * 4, mg = (size_t)rows * h->melb.Kpad * 4, mm = (size_t)rows * c.n_mels * 4;
	auto al = [](size_t x) { return (x + 255) / 256 * 256; };
	TTK_TRY(h->ws.reserve(al(fr) + al(sp) + al(mg) + al(mm)));
	float* frames = (float*)h->ws.p;
	float* spec = (float*)((char*)frames + al(fr));
	float* mag = (float*)((char*)spec + al(sp));
	float* mraw = (float*)((char*)mag + al(mg));
	{
		const int64_t total = rows * c.n_fft;
		hipLaunchKernelGGL(k_mel_frames, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wav, n, F, c.n_fft, c.hop, c.clip, total, frames);
	}
	GemmParams g = {};
	g.nseg = 1; g.seg[0] = {frames, c.n_fft, 0, 0};
	g.W = h->basis.w; g.ldw = h->basis.Kpad; g.M = (int)rows; g.N = h->basis.N; g.K = h->basis.Kpad; g.C = spec; g.ldc = h->basis.N; g.out_f32 = 1;
	launch_gemm(DT_F32, g, s);
	{
		const int64_t total = rows * h->melb.Kpad;
		hipLaunchKernelGGL(k_mel_mag, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, spec, h->nb, h->melb.Kpad, c.power, total, mag);
	}
	GemmParams g2 = {};
	g2.nseg = 1; g2.seg[0] = {mag, h->melb.Kpad, 0, 0};
	g2.W = h->melb.w; g2.ldw = h->melb.Kpad; g2.M = (int)rows; g2.N = c.n_mels; g2.K = h->melb.Kpad; g2.C = mraw; g2.ldc = c.n_mels; g2.out_f32 = 1;
	launch_gemm(DT_F32, g2, s);
	{
		const int64_t total = (int64_t)b * c.n_mels * F;
		hipLaunchKernelGGL(k_mel_log, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, mraw, F, c.n_mels, h->norms, total, mel);
	}
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

int ttk_resample_fir(const float* wav, int b, int n, const float* kernels, int gorig, int gnew, int width, float* out, int n_out, void* stream) {
	TTK_REQUIRE(wav && kernels && out, TTK_E_ARG, "ttk_resample_fir: null argument");
	TTK_REQUIRE(b >= 1 && n >= 1 && gorig >= 1 && gnew >= 1 && width >= 0 && n_out >= 1, TTK_E_ARG, "ttk_resample_fir: bad sizes (b=%d n=%d %d->%d width %d n_out=%d)",
				b, n, gorig, gnew, width, n_out);
	TTK_REQUIRE((int64_t)n_out <= ((int64_t)n / gorig + 1) * gnew, TTK_E_ARG, "ttk_resample_fir: n_out %d exceeds the %lld samples the clip yields", n_out,
				(long long)(((int64_t)n / gorig + 1) * gnew));
	const int64_t total = (int64_t)b * n_out;
	hipLaunchKernelGGL(k_resample_fir, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wav, n, kernels, gorig, gnew, width,
					   2 * width + gorig, n_out, total, out);
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

}  // extern "C"
