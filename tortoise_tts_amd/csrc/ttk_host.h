// Host-side plumbing shared by the two handles: error reporting, device memory, weight lookup/upload/packing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/ttk.h"
#include "ttk_kernels.h"

namespace ttk {

void set_error(const char* fmt, ...);

#define TTK_HIP(expr)                                                                                   \
	do {                                                                                                \
		hipError_t _e = (expr);                                                                         \
		if (_e != hipSuccess) {                                                                         \
			ttk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
			return TTK_E_HIP;                                                                           \
		}                                                                                               \
	} while (0)
#define TTK_TRY(expr)                \
	do {                             \
		int _r = (expr);             \
		if (_r != TTK_OK) return _r; \
	} while (0)
#define TTK_REQUIRE(cond, code, ...)     \
	do {                                 \
		if (!(cond)) {                   \
			ttk::set_error(__VA_ARGS__); \
			return code;                 \
		}                                \
	} while (0)

// owns every device allocation of a handle
struct Arena {
	std::vector<void*> ptrs;
	size_t bytes = 0;
	int alloc(void** out, size_t n) {
		if (n == 0) n = 16;
		TTK_HIP(hipMalloc(out, n));
		ptrs.push_back(*out);
		bytes += n;
		return TTK_OK;
	}
	void release() {
		for (void* p : ptrs) (void)hipFree(p);
		ptrs.clear();
	}
};

// grow-only workspace buffer
struct WsBuf {
	void* p = nullptr;
	size_t cap = 0;
	int reserve(size_t n) {
		if (n <= cap) return TTK_OK;
		if (p) TTK_HIP(hipFree(p));
		p = nullptr; cap = 0;
		TTK_HIP(hipMalloc(&p, n));
		cap = n;
		return TTK_OK;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct WeightMap {
	std::map<std::string, const ttk_weight_view*> m;
	WeightMap(const ttk_weight_view* w, int n) { for (int i = 0; i < n; ++i) m[w[i].name] = &w[i]; }
	const ttk_weight_view* find(const std::string& k) const { auto it = m.find(k); return it == m.end() ? nullptr : it->second; }
};

inline int64_t numel(const ttk_weight_view* v) { int64_t n = 1; for (int i = 0; i < v->ndim; ++i) n *= v->shape[i]; return n; }

// f32 tensor copied to the device as is (norm scales, biases, embedding tables)
int upload_f32(Arena& ar, const WeightMap& wm, const std::string& name, int64_t expect_numel, float** out);

// a GEMM operand: T-typed [ntap][Npad][Kpad] (+ optional MFMA-fragment copy) and its f32 bias
struct Mat {
	void* w = nullptr;      // [ntap][Npad][Kpad]
	void* wfrag = nullptr;  // [Npad/16][Kpad/32][64][8], decode path only (one byte per element when w8)
	bool w8 = false;        // DT_FP8W: weights rounded to fp8-e4m3 * wscale; `w` holds them exactly in bf16, `wfrag` as fp8 bytes
	float wscale = 1.f;
	int wes = 2;            // bytes per element of `w` (1: DT_FP8, the dense GEMM reads fp8 bytes and applies wscale in its epilogue)
	float* bias = nullptr;
	int N = 0, K = 0, Npad = 0, Kpad = 0, ntap = 1;
	// decode path with the preceding LayerNorm folded in (fold_layernorm): fragment-order gamma o W, its column sums, b + beta W
	void* wfrag_fold = nullptr;
	float *csum = nullptr, *bias_fold = nullptr;
};
// layout: PK_NK / PK_KN / PK_CONV3 ; N, K are the logical sizes checked against the tensor
int upload_mat(Arena& ar, const WeightMap& wm, int dt, const std::string& wname, const std::string& bname, int layout,
			   int N, int K, bool frag, Mat* out, int ntap = 0);   // ntap: kernel size for PK_CONVK / PK_CONVT

// For a PK_KN matrix uploaded with a fragment copy: build the folded-LayerNorm operands of the decode launch (see SkinnyParams.g1) from the
// LayerNorm's gamma / beta.  fp8 weights: folded AFTER the rounding, the folded operand held in the kernel type (see the definition).
int fold_layernorm(Arena& ar, const WeightMap& wm, int dt, const std::string& wname, const std::string& bname, const std::string& gname,
				   const std::string& betaname, Mat* m);

// sample.hip: the fused per-token sampling launch; emb / pos / x_out non-null = also write the next decode step's input rows
int launch_sample_step(const ttk_sample_args* a, const float* emb, const float* pos, float* x_out, int d, int pos_rows, void* x_frag, int x_frag_f32,
					   hipStream_t stream, const char* who);

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace ttk
