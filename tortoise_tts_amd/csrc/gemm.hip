// Dense "NT" GEMM on MFMA for gfx950:  C[M,N] = epi( sum_seg shift(A_seg)[M,K] * W_seg[N,K]^T ).
//
// Serves every GEMM-shaped op of both networks in their channels-last / token-major internal layout:
//   GPT-2 c_attn / c_proj / c_fc / mlp.c_proj (HF Conv1D, HF:pytorch_utils.py:95-120) for prefill + latent pass,
//   mel_head, DiffusionTTS 1x1 convs (qkv, proj_out, in_layers.2, integrating_conv = 2 concatenated segments),
//   k=3 convs (inp_block, latent_conditioner.0, out_layers.3, out.2 = 3 row-shifted segments, zero padded at the
//   edges of each batch element), emb_layers / time_embed linears.   (/root/reference/tortoise_tts/models/diffusion.py:1316-1376,1517-1574)
//
// Tiling: 256 threads = 4 waves (2x2), BMxBN block tile, 128-byte-row LDS tiles (64 bf16 / 32 f32 of K),
// XOR-swizzled 16-byte chunks, double-buffered LDS with register-staged prefetch (global loads of tile k+1 in
// flight under the MFMAs of tile k; one barrier per tile), 16x16 MFMA sub-tiles, f32 accumulate.
// Roofline: MFMA-bound for the diffusion shapes (M = b*T ~ 2k rows, N,K in 1k..3k); bytes/flop is tiny.
#include "ttk_common.h"
#include "ttk_kernels.h"

namespace ttk {

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void k_gemm(GemmParams p) {
	constexpr int ES = sizeof(T);
	constexpr int BKE = 128 / ES;      // K elements per tile row
	constexpr int KSTEPS = BKE / 32;   // MFMA k-steps per tile
	constexpr int FCH = 8 * ES / 16;   // 16-byte chunks per fragment
	constexpr int EPC = 16 / ES;       // elements per chunk
	constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
	constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;
	typedef typename Frag<T>::type FragT;
	extern __shared__ __attribute__((aligned(16))) char smem[];

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1;
	const int tiles_m = (p.M + BM - 1) / BM;
	const int m0 = (blockIdx.x % tiles_m) * BM, n0 = (blockIdx.x / tiles_m) * BN;
	const int KT = p.K / BKE;
	const int NTILES = p.nseg * KT;

	uint4 ra[A_CH], rb[B_CH];
	int trow[A_CH];   // row index inside its batch element (for shifted segments)
#pragma unroll
	for (int i = 0; i < A_CH; ++i) {
		const int gm = m0 + ((tid + 256 * i) >> 3);
		trow[i] = p.rows_per_batch > 0 ? gm % p.rows_per_batch : 0;
	}

	auto load_tile = [&](int kt) {
		const int sg = kt / KT;
		const int k0 = (kt - sg * KT) * BKE;
		const T* Ab = (const T*)p.seg[sg].A;
		const int64_t lda = p.seg[sg].lda;
		const int shift = p.seg[sg].shift;
		const T* Wb = (const T*)p.W + p.seg[sg].w_off;
#pragma unroll
		for (int i = 0; i < A_CH; ++i) {
			const int id = tid + 256 * i, row = id >> 3, c = id & 7;
			const int gm = m0 + row;
			bool ok = gm < p.M;
			if (shift != 0) { const int t = trow[i] + shift; ok = ok && t >= 0 && t < p.rows_per_batch; }
			ra[i] = ok ? *(const uint4*)(Ab + (int64_t)(gm + shift) * lda + k0 + c * EPC) : make_uint4(0, 0, 0, 0);
		}
#pragma unroll
		for (int i = 0; i < B_CH; ++i) {
			const int id = tid + 256 * i, row = id >> 3, c = id & 7;
			rb[i] = *(const uint4*)(Wb + (int64_t)(n0 + row) * p.ldw + k0 + c * EPC);
		}
	};
	auto store_tile = [&](int buf) {
		char* As = smem + buf * (BM + BN) * 128;
		char* Bs = As + BM * 128;
#pragma unroll
		for (int i = 0; i < A_CH; ++i) {
			const int id = tid + 256 * i, row = id >> 3, c = id & 7;
			*(uint4*)(As + row * 128 + ((c ^ (row & 7)) << 4)) = ra[i];
		}
#pragma unroll
		for (int i = 0; i < B_CH; ++i) {
			const int id = tid + 256 * i, row = id >> 3, c = id & 7;
			*(uint4*)(Bs + row * 128 + ((c ^ (row & 7)) << 4)) = rb[i];
		}
	};

	f32x4 acc[MI][NI];
#pragma unroll
	for (int i = 0; i < MI; ++i)
#pragma unroll
		for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

	auto compute = [&](int buf) {
		const char* As = smem + buf * (BM + BN) * 128;
		const char* Bs = As + BM * 128;
#pragma unroll
		for (int ks = 0; ks < KSTEPS; ++ks) {
			union { FragT v; uint4 q[FCH]; } a[MI], b[NI];
			const int c0 = (ks * 32 + 8 * (lane >> 4)) / EPC;
#pragma unroll
			for (int i = 0; i < MI; ++i) {
				const int row = wm * WM + 16 * i + (lane & 15);
#pragma unroll
				for (int f = 0; f < FCH; ++f) a[i].q[f] = *(const uint4*)(As + row * 128 + (((c0 + f) ^ (row & 7)) << 4));
			}
#pragma unroll
			for (int j = 0; j < NI; ++j) {
				const int row = wn * WN + 16 * j + (lane & 15);
#pragma unroll
				for (int f = 0; f < FCH; ++f) b[j].q[f] = *(const uint4*)(Bs + row * 128 + (((c0 + f) ^ (row & 7)) << 4));
			}
#pragma unroll
			for (int i = 0; i < MI; ++i)
#pragma unroll
				for (int j = 0; j < NI; ++j) acc[i][j] = mma16<T>(a[i].v, b[j].v, acc[i][j]);
		}
	};

	load_tile(0);
	store_tile(0);
	__syncthreads();
	for (int kt = 0; kt < NTILES; ++kt) {
		const int buf = kt & 1;
		if (kt + 1 < NTILES) load_tile(kt + 1);
		compute(buf);
		if (kt + 1 < NTILES) store_tile(buf ^ 1);
		__syncthreads();
	}

	// epilogue: accumulator (i, j) register r is row 16i + 4*(lane>>4) + r, column 16j + (lane&15)
#pragma unroll
	for (int i = 0; i < MI; ++i) {
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int gm = m0 + wm * WM + 16 * i + 4 * (lane >> 4) + r;
			if (gm >= p.M) continue;
#pragma unroll
			for (int j = 0; j < NI; ++j) {
				const int gn = n0 + wn * WN + 16 * j + (lane & 15);
				if (gn >= p.N) continue;
				float v = acc[i][j][r];
				if (p.bias) v += p.bias[gn];
				v = apply_act(v, p.act);
				if (p.residual) v += p.residual[(int64_t)gm * p.ldr + gn];
				if (p.transpose_out) {
					const int bb = gm / p.rows_per_batch, t = gm - bb * p.rows_per_batch;
					((float*)p.C)[((int64_t)bb * p.N + gn) * p.rows_per_batch + t] = v;
				} else if (p.out_f32) {
					((float*)p.C)[(int64_t)gm * p.ldc + gn] = v;
				} else {
					((T*)p.C)[(int64_t)gm * p.ldc + gn] = cvt<T>(v);
				}
			}
		}
	}
}

template <typename T>
static void launch_gemm_t(const GemmParams& p, hipStream_t s) {
	const int t128 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
	if (t128 >= 192) {
		const int grid = t128;
		hipLaunchKernelGGL((k_gemm<T, 128, 128>), dim3(grid), dim3(256), 2 * 256 * 128, s, p);
	} else {
		const int grid = ((p.M + 63) / 64) * ((p.N + 63) / 64);
		hipLaunchKernelGGL((k_gemm<T, 64, 64>), dim3(grid), dim3(256), 2 * 128 * 128, s, p);
	}
}

void launch_gemm(int dt, const GemmParams& p, hipStream_t s) {
	ProfScope prof(PROF_GEMM, 2.0 * p.M * p.N * (double)p.K * p.nseg, s);
	if (dt == DT_BF16) launch_gemm_t<bf16>(p, s);
	else launch_gemm_t<float>(p, s);
}

}  // namespace ttk
