// Small HBM/L2-bound helpers: embedding gathers, layout changes at the [b,C,T] boundary, timestep embedding,
// and the fused per-step diffusion epilogue (guidance mix + x0 clamp + DDIM / ancestral update).
#include "ttk_common.h"
#include "ttk_kernels.h"

namespace ttk {

__global__ void k_set_int(int* p, int v) { *p = v; }
__global__ void k_add_int(int* p, int v) { *p += v; }
void launch_set_int(int* p, int v, hipStream_t s) { hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, s, p, v); }
void launch_add_int(int* p, int v, hipStream_t s) { hipLaunchKernelGGL(k_add_int, dim3(1), dim3(1), 0, s, p, v); }
__global__ void k_fill_int(int* p, int v, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
void launch_fill_int(int* p, int v, int n, hipStream_t s) { if (n > 0) hipLaunchKernelGGL(k_fill_int, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n); }
__global__ void k_fill_int2(int* p, int a, int b, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { p[2 * i] = a; p[2 * i + 1] = b; } }
void launch_fill_int2(int* p, int a, int b, int n, hipStream_t s) { if (n > 0) hipLaunchKernelGGL(k_fill_int2, dim3((n + 255) / 256), dim3(256), 0, s, p, a, b, n); }

// out[r] = A[ia[r]] + Bt[ib[r]]      (unified_voice.py:582,590,641: embedding + learned position embedding)
__global__ void k_gather_add(const float* A, const int* ia, const float* Bt, const int* ib, float* out, int rows, int d) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int d4 = d / 4;
	if (idx >= (int64_t)rows * d4) return;
	const int r = (int)(idx / d4), c = (int)(idx - (int64_t)r * d4) * 4;
	float4 v = *(const float4*)(A + (int64_t)ia[r] * d + c);
	if (Bt) { const float4 w = *(const float4*)(Bt + (int64_t)ib[r] * d + c); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
	*(float4*)(out + (int64_t)r * d + c) = v;
}
void launch_gather_add(const float* A, const int* ia, const float* Bt, const int* ib, float* out, int rows, int d, hipStream_t s) {
	const int64_t total = (int64_t)rows * (d / 4);
	hipLaunchKernelGGL(k_gather_add, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, A, ia, Bt, ib, out, rows, d);
}

// decode: x[b] = mel_embedding[tok[b]] + mel_pos_embedding[*d_pos + pos_off]
// (unified_voice.py:213-214; with d_pos = cache length before this step = P + k, pos_off = 1 - P gives the k + 1 quirk)
// frag (optional): the same rows, T-typed, in A-fragment order [m_tile][d/32][lane][8] for a folded-LayerNorm launch (skinny.hip)
__global__ void k_decode_embed(const float* emb, const int64_t* tok, const float* pos, const int* d_pos, int pos_off, int pos_rows, float* out, int B, int d,
							   void* frag, int frag_f32) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	const int d4 = d / 4;
	if (idx >= B * d4) return;
	const int b = idx / d4, c = (idx - b * d4) * 4;
	int pi = *d_pos + pos_off;
	pi = pi < 0 ? 0 : (pi >= pos_rows ? pos_rows - 1 : pi);   // guard; the host validates lengths up front
	const float4 e = *(const float4*)(emb + tok[b] * d + c), w = *(const float4*)(pos + (int64_t)pi * d + c);
	const float4 v = make_float4(e.x + w.x, e.y + w.y, e.z + w.z, e.w + w.w);
	*(float4*)(out + (int64_t)b * d + c) = v;
	if (frag) {
		const int64_t fi = ((((int64_t)(b >> 4) * (d / 32) + (c >> 5)) * 64 + ((c >> 3) & 3) * 16 + (b & 15)) * 8 + (c & 7));
		store4_kind(frag, fi, v, frag_f32);      // frag_f32: ttk::ElemKind of the copy
	}
}
void launch_decode_embed(const float* emb, const int64_t* tok, const float* pos, const int* d_pos, int pos_off, int pos_rows, float* out, int B, int d, hipStream_t s,
						 void* frag, int frag_f32) {
	hipLaunchKernelGGL(k_decode_embed, dim3((B * (d / 4) + 255) / 256), dim3(256), 0, s, emb, tok, pos, d_pos, pos_off, pos_rows, out, B, d, frag, frag_f32);
}

__global__ void k_copy_rows(const float* src, int64_t lds_, float* dst, int64_t ldd, int rows, int d) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int d4 = d / 4;
	if (idx >= (int64_t)rows * d4) return;
	const int r = (int)(idx / d4), c = (int)(idx - (int64_t)r * d4) * 4;
	*(float4*)(dst + r * ldd + c) = *(const float4*)(src + r * lds_ + c);
}
void launch_copy_rows(const float* src, int64_t lds_, float* dst, int64_t ldd, int rows, int d, hipStream_t s) {
	const int64_t total = (int64_t)rows * (d / 4);
	hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, lds_, dst, ldd, rows, d);
}

template <typename T>
__global__ void k_cast(const float* src, T* dst, int64_t n) {
	const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
	if (i + 3 < n) {
		const float4 v = *(const float4*)(src + i);
		dst[i] = cvt<T>(v.x); dst[i + 1] = cvt<T>(v.y); dst[i + 2] = cvt<T>(v.z); dst[i + 3] = cvt<T>(v.w);
	} else {
		for (int64_t j = i; j < n; ++j) dst[j] = cvt<T>(src[j]);
	}
}
void launch_cast(int dt, const float* src, void* dst, int64_t n, hipStream_t s) {
	const unsigned grid = (unsigned)((n / 4 + 256) / 256);
	if (dt == DT_BF16) hipLaunchKernelGGL((k_cast<bf16>), dim3(grid), dim3(256), 0, s, src, (bf16*)dst, n);
	else if (dt == DT_F16) hipLaunchKernelGGL((k_cast<f16>), dim3(grid), dim3(256), 0, s, src, (f16*)dst, n);
	else hipLaunchKernelGGL((k_cast<float>), dim3(grid), dim3(256), 0, s, src, (float*)dst, n);
}

// channel-first f32 [nb][C][T] -> channels-last T-typed [rep*nb*T][ldo], zero padded columns C..ldo (LDS tile transpose)
template <typename T>
__global__ void k_cf_to_cl(const float* src, int nb, int C, int Tn, T* dst, int64_t ldo, int rep, const int* tlen) {
	__shared__ float tile[32][33];
	const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
	const int tl = tlen ? tlen[b] : Tn;      // ragged batch: frames [tl, Tn) of element b are padding and become ZERO rows (the k = 3 convs' edge)
	for (int i = ty; i < 32; i += 8) {
		const int c = c0 + i, t = t0 + tx;
		tile[i][tx] = (c < C && t < tl) ? src[((int64_t)b * C + c) * Tn + t] : 0.f;
	}
	__syncthreads();
	for (int i = ty; i < 32; i += 8) {
		const int t = t0 + i, c = c0 + tx;
		if (t < Tn && c < ldo)
			for (int r = 0; r < rep; ++r) dst[(((int64_t)r * nb + b) * Tn + t) * ldo + c] = cvt<T>(tile[tx][i]);
	}
}
void launch_cf_to_cl(int dt, const float* src, int nb, int C, int T, void* dst, int64_t ldo, int rep, hipStream_t s, const int* tlen) {
	dim3 grid((T + 31) / 32, (unsigned)((ldo + 31) / 32), nb);
	if (dt == DT_BF16) hipLaunchKernelGGL((k_cf_to_cl<bf16>), grid, dim3(256), 0, s, src, nb, C, T, (bf16*)dst, ldo, rep, tlen);
	else if (dt == DT_F16) hipLaunchKernelGGL((k_cf_to_cl<f16>), grid, dim3(256), 0, s, src, nb, C, T, (f16*)dst, ldo, rep, tlen);
	else hipLaunchKernelGGL((k_cf_to_cl<float>), grid, dim3(256), 0, s, src, nb, C, T, (float*)dst, ldo, rep, tlen);
}

__global__ void k_cl_to_cf(const float* src, int nb, int C, int Tn, float* dst) {
	__shared__ float tile[32][33];
	const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
	for (int i = ty; i < 32; i += 8) {
		const int t = t0 + i, c = c0 + tx;
		tile[i][tx] = (c < C && t < Tn) ? src[((int64_t)b * Tn + t) * C + c] : 0.f;
	}
	__syncthreads();
	for (int i = ty; i < 32; i += 8) {
		const int c = c0 + i, t = t0 + tx;
		if (c < C && t < Tn) dst[((int64_t)b * C + c) * Tn + t] = tile[tx][i];
	}
}
void launch_cl_to_cf(const float* src, int nb, int C, int T, float* dst, hipStream_t s) {
	dim3 grid((T + 31) / 32, (C + 31) / 32, nb);
	hipLaunchKernelGGL(k_cl_to_cf, grid, dim3(256), 0, s, src, nb, C, T, dst);
}

template <typename T>
__global__ void k_bcast_rows(const float* vec, int rows, int C, T* dst) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (int64_t)rows * C) return;
	dst[idx] = cvt<T>(vec[idx % C]);
}
void launch_bcast_rows(int dt, const float* vec, int rows, int C, void* dst, hipStream_t s) {
	const int64_t total = (int64_t)rows * C;
	const unsigned grid = (unsigned)((total + 255) / 256);
	if (dt == DT_BF16) hipLaunchKernelGGL((k_bcast_rows<bf16>), dim3(grid), dim3(256), 0, s, vec, rows, C, (bf16*)dst);
	else if (dt == DT_F16) hipLaunchKernelGGL((k_bcast_rows<f16>), dim3(grid), dim3(256), 0, s, vec, rows, C, (f16*)dst);
	else hipLaunchKernelGGL((k_bcast_rows<float>), dim3(grid), dim3(256), 0, s, vec, rows, C, (float*)dst);
}

// diffusion.py:1277-1295: emb[i] = cat(cos(t*f), sin(t*f)); f_j = exp(-ln(1e4) * j / half) is a host-computed f32 table
// (a 1-ulp difference in f_j is a 2e-4 rad phase error at t = 3999, so the table is built once by the same f32 ops
// as the reference and only cos/sin run here)
template <typename T>
__global__ void k_timestep_embedding(const int64_t* t, int64_t t_host, int n, int C, const float* freqs, T* out) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	const int half = C / 2;
	if (idx >= n * half) return;
	const int i = idx / half, j = idx - i * half;
	const float tv = (float)(t ? t[i] : t_host);
	const float a = tv * freqs[j];
	out[(int64_t)i * C + j] = cvt<T>(cosf(a));
	out[(int64_t)i * C + half + j] = cvt<T>(sinf(a));
}
void launch_timestep_embedding(int dt, const int64_t* t, int64_t t_host, int n, int C, const float* freqs, void* out, hipStream_t s) {
	const unsigned grid = (n * (C / 2) + 255) / 256;
	if (dt == DT_BF16) hipLaunchKernelGGL((k_timestep_embedding<bf16>), dim3(grid), dim3(256), 0, s, t, t_host, n, C, freqs, (bf16*)out);
	else if (dt == DT_F16) hipLaunchKernelGGL((k_timestep_embedding<f16>), dim3(grid), dim3(256), 0, s, t, t_host, n, C, freqs, (f16*)out);
	else hipLaunchKernelGGL((k_timestep_embedding<float>), dim3(grid), dim3(256), 0, s, t, t_host, n, C, freqs, (float*)out);
}

template <typename T>
__global__ void k_silu_cast(const float* x, T* y, int64_t n) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) y[i] = cvt<T>(silu_precise(x[i]));
}
void launch_silu_cast(int dt, const float* x, void* y, int64_t n, hipStream_t s) {
	const unsigned grid = (unsigned)((n + 255) / 256);
	if (dt == DT_BF16) hipLaunchKernelGGL((k_silu_cast<bf16>), dim3(grid), dim3(256), 0, s, x, (bf16*)y, n);
	else if (dt == DT_F16) hipLaunchKernelGGL((k_silu_cast<f16>), dim3(grid), dim3(256), 0, s, x, (f16*)y, n);
	else hipLaunchKernelGGL((k_silu_cast<float>), dim3(grid), dim3(256), 0, s, x, (float*)y, n);
}

// Per-step sampler epilogue on channel-first buffers.  out_c / out_u: network outputs [nb][2C][T] (eps | var) of the
// conditioned / unconditioned evaluation; x [nb][C][T] updated in place.
//   guidance mix        diffusion.py:390-396   eps = (1 + cfk) eps_c - cfk eps_u
//   x0 prediction       diffusion.py:433-438, clamp :401-403
//   DDIM (eta = 0)      diffusion.py:675-693   eps' = (sqrt_recip_ac x - x0) / sqrt_recipm1_ac ; x = x0 sqrt(ac_prev) + sqrt(1-ac_prev) eps'
//   ancestral "p"       diffusion.py:301-323, 366-373, 545-553   mean = c1 x0 + c2 x ; log_var from the learned range
// xcl (round 6, optional): the NEXT step's channels-last operand copy of the updated x -- what k_cf_to_cl would make of it ([rep * nb * T][ldo] in the kernel type, element kind
// `ekind`; the zero padding columns C .. ldo stay as the first step's k_cf_to_cl left them) -- so that a loop of steps needs that launch once, not once per step
__global__ void k_diffusion_step(const float* out_c, const float* out_u, float* x, const float* noise, int nb, int C, int Tn, StepCoefs k, void* xcl, int ldo, int rep, int ekind) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t per = (int64_t)C * Tn;
	if (idx >= nb * per) return;
	const int b = (int)(idx / per);
	const int64_t r = idx - b * per;
	float eps = out_c[(int64_t)b * 2 * per + r];
	if (k.cfk >= 0.f) {
		const float eu = out_u[(int64_t)b * 2 * per + r];
		eps = (1.0f + k.cfk) * eps - k.cfk * eu;
	}
	const float xv = x[idx];
	float x0 = k.sqrt_recip_ac * xv - k.sqrt_recipm1_ac * eps;
	x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
	float xn;
	if (k.sampler == 0) {
		const float e2 = (k.sqrt_recip_ac * xv - x0) / k.sqrt_recipm1_ac;
		xn = x0 * k.sqrt_ac_prev + k.sqrt_1m_ac_prev * e2;
	} else {
		const float var = out_c[(int64_t)b * 2 * per + per + r];
		const float frac = (var + 1.0f) / 2.0f;
		const float log_var = frac * k.max_log + (1.0f - frac) * k.min_log;
		const float mean = k.coef1 * x0 + k.coef2 * xv;
		xn = mean + (k.nonzero ? expf(0.5f * log_var) * noise[idx] : 0.f);
	}
	x[idx] = xn;
	if (xcl) {
		const int c = (int)(r / Tn), t = (int)(r - (int64_t)c * Tn);
		for (int q = 0; q < rep; ++q) {
			const int64_t o = (((int64_t)q * nb + b) * Tn + t) * ldo + c;
			if (ekind == EK_F32) ((float*)xcl)[o] = xn;
			else if (ekind == EK_F16) ((f16*)xcl)[o] = cvt<f16>(xn);
			else ((bf16*)xcl)[o] = cvt<bf16>(xn);
		}
	}
}
void launch_diffusion_step(const float* out_c, const float* out_u, float* x, const float* noise, int nb, int C, int T, StepCoefs c, hipStream_t s, void* xcl, int ldo, int rep, int ekind) {
	const int64_t total = (int64_t)nb * C * T;
	hipLaunchKernelGGL(k_diffusion_step, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out_c, out_u, x, noise, nb, C, T, c, xcl, ldo, rep, ekind);
}

}  // namespace ttk
