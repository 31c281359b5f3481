#include "ttk_host.h"

#include <stdarg.h>

namespace ttk {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}
const char* get_error() { return g_err; }

int upload_f32(Arena& ar, const WeightMap& wm, const std::string& name, int64_t expect_numel, float** out) {
	const ttk_weight_view* v = wm.find(name);
	TTK_REQUIRE(v != nullptr, TTK_E_WEIGHT, "missing weight '%s'", name.c_str());
	TTK_REQUIRE(numel(v) == expect_numel, TTK_E_WEIGHT, "weight '%s' has %lld elements, expected %lld", name.c_str(),
				(long long)numel(v), (long long)expect_numel);
	TTK_TRY(ar.alloc((void**)out, (size_t)expect_numel * sizeof(float)));
	TTK_HIP(hipMemcpy(*out, v->data, (size_t)expect_numel * sizeof(float), hipMemcpyDefault));
	return TTK_OK;
}

int upload_mat(Arena& ar, const WeightMap& wm, int dt, const std::string& wname, const std::string& bname, int layout,
			   int N, int K, bool frag, Mat* out, int ntap_in) {
	const ttk_weight_view* v = wm.find(wname);
	TTK_REQUIRE(v != nullptr, TTK_E_WEIGHT, "missing weight '%s'", wname.c_str());
	const int ntap = ntap_in > 0 ? ntap_in : (layout == PK_CONV3 ? 3 : 1);
	TTK_REQUIRE(numel(v) == (int64_t)N * K * ntap, TTK_E_WEIGHT, "weight '%s' has %lld elements, expected %lld", wname.c_str(),
				(long long)numel(v), (long long)N * K * ntap);
	out->N = N; out->K = K; out->ntap = ntap;
	out->Npad = round_up(N, 128);
	out->Kpad = round_up(K, dt == DT_FP8 ? 128 : 64);
	const bool w8 = dt == DT_FP8W || dt == DT_FP8;
	const int kdt = kernel_dtype(dt);
	const size_t es = dt == DT_FP8 ? 1 : dtype_size(kdt);
	out->wes = (int)es;
	float* tmp = nullptr;
	TTK_HIP(hipMalloc((void**)&tmp, (size_t)numel(v) * sizeof(float)));
	hipError_t e = hipMemcpy(tmp, v->data, (size_t)numel(v) * sizeof(float), hipMemcpyDefault);
	if (e != hipSuccess) { (void)hipFree(tmp); set_error("hipMemcpy of '%s' failed: %s", wname.c_str(), hipGetErrorString(e)); return TTK_E_HIP; }
	if (w8) {   // round the weights to the fp8 grid first: every copy made below then holds the same values
		float amax = 0.f;
		if (device_absmax(tmp, numel(v), &amax) != 0) { (void)hipFree(tmp); set_error("absmax of '%s' failed", wname.c_str()); return TTK_E_HIP; }
		out->w8 = true;
		out->wscale = fp8_scale_for(amax);
		launch_fp8_roundtrip(tmp, numel(v), out->wscale, 0);
	}
	const size_t wbytes = (size_t)ntap * out->Npad * out->Kpad * es;
	int rc = ar.alloc(&out->w, wbytes);
	if (rc == TTK_OK) {
		if (dt == DT_FP8) launch_pack_nk_f8(tmp, layout, N, K, out->Npad, out->Kpad, out->wscale, out->w, 0, ntap);
		else launch_pack_nk(kdt, tmp, layout, N, K, out->Npad, out->Kpad, out->w, 0, ntap);
		if (frag && ntap == 1 && dt != DT_FP8) {
			rc = ar.alloc(&out->wfrag, w8 ? wbytes / 2 : wbytes);
			if (rc == TTK_OK) {
				if (w8) launch_pack_frag_fp8(out->w, out->Npad, out->Kpad, out->wscale, out->wfrag, 0);
				else launch_pack_frag(kdt, out->w, out->Npad, out->Kpad, out->wfrag, 0);
			}
		}
	}
	e = hipDeviceSynchronize();
	(void)hipFree(tmp);
	if (rc != TTK_OK) return rc;
	if (e != hipSuccess) { set_error("weight packing of '%s' failed: %s", wname.c_str(), hipGetErrorString(e)); return TTK_E_HIP; }
	if (!bname.empty()) TTK_TRY(upload_f32(ar, wm, bname, N, &out->bias));
	return TTK_OK;
}

int fold_layernorm(Arena& ar, const WeightMap& wm, int dt, const std::string& wname, const std::string& bname, const std::string& gname,
				   const std::string& betaname, Mat* m) {
	TTK_REQUIRE(m->wfrag && m->ntap == 1, TTK_E_ARG, "fold_layernorm('%s'): needs a fragment-order matrix", wname.c_str());
	const ttk_weight_view *vw = wm.find(wname), *vb = wm.find(bname), *vg = wm.find(gname), *vbe = wm.find(betaname);
	TTK_REQUIRE(vw && vb && vg && vbe, TTK_E_WEIGHT, "fold_layernorm: missing one of '%s', '%s', '%s', '%s'", wname.c_str(), bname.c_str(), gname.c_str(), betaname.c_str());
	const int N = m->N, K = m->K;
	TTK_REQUIRE(numel(vw) == (int64_t)N * K && numel(vb) == N && numel(vg) == K && numel(vbe) == K, TTK_E_WEIGHT, "fold_layernorm('%s'): shape mismatch", wname.c_str());
	const size_t es = dtype_size(dt);
	float *w = nullptr, *g = nullptr, *be = nullptr, *b = nullptr;
	void* wt = nullptr;
	auto cleanup = [&]() { for (void* p : {(void*)w, (void*)g, (void*)be, (void*)b, wt}) if (p) (void)hipFree(p); };
	hipError_t e = hipMalloc((void**)&w, (size_t)N * K * 4);
	if (e == hipSuccess) e = hipMalloc((void**)&g, (size_t)K * 4);
	if (e == hipSuccess) e = hipMalloc((void**)&be, (size_t)K * 4);
	if (e == hipSuccess) e = hipMalloc((void**)&b, (size_t)N * 4);
	if (e == hipSuccess) e = hipMalloc(&wt, (size_t)m->Npad * m->Kpad * es);
	if (e == hipSuccess) e = hipMemcpy(w, vw->data, (size_t)N * K * 4, hipMemcpyDefault);
	if (e == hipSuccess) e = hipMemcpy(g, vg->data, (size_t)K * 4, hipMemcpyDefault);
	if (e == hipSuccess) e = hipMemcpy(be, vbe->data, (size_t)K * 4, hipMemcpyDefault);
	if (e == hipSuccess) e = hipMemcpy(b, vb->data, (size_t)N * 4, hipMemcpyDefault);
	if (e != hipSuccess) { cleanup(); set_error("fold_layernorm('%s'): %s", wname.c_str(), hipGetErrorString(e)); return TTK_E_HIP; }
	int rc = ar.alloc(&m->wfrag_fold, (size_t)m->Npad * m->Kpad * es);
	if (rc == TTK_OK) rc = ar.alloc((void**)&m->csum, (size_t)N * 4);
	if (rc == TTK_OK) rc = ar.alloc((void**)&m->bias_fold, (size_t)N * 4);
	if (rc != TTK_OK) { cleanup(); return rc; }
	// fp8 weights: the mode is defined on the reference's matrices, so those are rounded first (same scale as upload_mat) and the fold is taken
	// of the ROUNDED matrix, in the kernel arithmetic -- what a bf16 handle built from the rounded weights computes, bit for bit.  The folded
	// operand is a bf16 matrix (gamma o W^ is not on the fp8 grid): these two launches stream 2 bytes per weight, the decode step being bound by
	// its launch chain, not by weight bytes
	if (m->w8) launch_fp8_roundtrip(w, (int64_t)N * K, m->wscale, 0);
	launch_bias_fold(w, be, b, K, N, m->bias_fold, 0);               // from the unscaled weights
	launch_scale_kn(w, g, K, N, 0);
	launch_pack_nk(dt, w, PK_KN, N, K, m->Npad, m->Kpad, wt, 0, 1);
	launch_pack_frag(dt, wt, m->Npad, m->Kpad, m->wfrag_fold, 0);
	launch_rowsum(dt, wt, m->Kpad, N, K, m->csum, 0);
	e = hipDeviceSynchronize();
	cleanup();
	if (e != hipSuccess) { set_error("fold_layernorm('%s'): %s", wname.c_str(), hipGetErrorString(e)); return TTK_E_HIP; }
	return TTK_OK;
}

bool g_prof_on = false;
namespace {
struct ProfRec { hipEvent_t a, b; int kind; double work; };
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_used = 0;
hipEvent_t pool_get() {
	if (g_pool_used == g_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); g_pool.push_back(e); }
	return g_pool[g_pool_used++];
}
}  // namespace
void prof_start(int kind, double work, hipStream_t s) {
	ProfRec r; r.a = pool_get(); r.b = pool_get(); r.kind = kind; r.work = work;
	(void)hipEventRecord(r.a, s);
	g_recs.push_back(r);
}
void prof_stop(hipStream_t s) { (void)hipEventRecord(g_recs.back().b, s); }
void prof_pair(int kind, double work, hipEvent_t* start, hipEvent_t* stop) {
	ProfRec r; r.a = pool_get(); r.b = pool_get(); r.kind = kind; r.work = work;
	g_recs.push_back(r);
	*start = r.a; *stop = r.b;
}

}  // namespace ttk

extern "C" {
int ttk_prof_begin(void) {
	ttk::g_recs.clear();
	ttk::g_pool_used = 0;
	ttk::g_prof_on = true;
	return TTK_OK;
}
int ttk_prof_end(ttk_prof_result* out, int n_kinds) {
	ttk::g_prof_on = false;
	TTK_REQUIRE(out && n_kinds >= ttk::PROF_KINDS, TTK_E_ARG, "ttk_prof_end: need room for %d kinds", (int)ttk::PROF_KINDS);
	TTK_HIP(hipDeviceSynchronize());
	for (int i = 0; i < n_kinds; ++i) { out[i].ms = 0; out[i].launches = 0; out[i].work = 0; }
	for (const auto& r : ttk::g_recs) {
		float ms = 0.f;
		TTK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
		out[r.kind].ms += ms; out[r.kind].launches += 1; out[r.kind].work += r.work;
	}
	ttk::g_recs.clear();
	return TTK_OK;
}
int ttk_version(void) { return TTK_VERSION; }
const char* ttk_last_error(void) { return ttk::get_error(); }
}

extern "C" int ttk_fp8_round_weights(float* x, int64_t n, float* scale_out, void* stream) {
	using namespace ttk;
	TTK_REQUIRE(x && n > 0, TTK_E_ARG, "ttk_fp8_round_weights: null or empty array");
	TTK_HIP(hipStreamSynchronize((hipStream_t)stream));
	float amax = 0.f;
	TTK_REQUIRE(device_absmax(x, n, &amax) == 0, TTK_E_HIP, "ttk_fp8_round_weights: absmax reduction failed");
	const float s = fp8_scale_for(amax);
	launch_fp8_roundtrip(x, n, s, (hipStream_t)stream);
	TTK_HIP(hipGetLastError());
	if (scale_out) *scale_out = s;
	return TTK_OK;
}

// The dense NT GEMM on caller-provided operands: kernel-level numerics tests (a plain f32 matmul of the same operands is the reference) and tuning.
extern "C" int ttk_gemm_nt(int dtype, const void* A, const void* W, int M, int N, int K, float out_scale, const float* bias, float* C, void* stream) {
	using namespace ttk;
	TTK_REQUIRE(A && W && C, TTK_E_ARG, "ttk_gemm_nt: null argument");
	TTK_REQUIRE(dtype == TTK_F32 || dtype == TTK_BF16 || dtype == TTK_F16 || dtype == TTK_FP8, TTK_E_ARG, "ttk_gemm_nt: dtype must be TTK_F32, TTK_BF16, TTK_F16 or TTK_FP8 (fp8-e4m3 bytes), got %d", dtype);
	const int kmul = dtype == TTK_FP8 ? 128 : (dtype == TTK_F32 ? 32 : 64);
	TTK_REQUIRE(M >= 1 && N >= 128 && N % 128 == 0 && K >= kmul && K % kmul == 0, TTK_E_ARG,
				"ttk_gemm_nt: need M >= 1, N %% 128 == 0, K %% %d == 0 (got M=%d N=%d K=%d)", kmul, M, N, K);
	GemmParams g = {};
	g.nseg = 1; g.seg[0] = {A, K, 0, 0};
	g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K; g.bias = bias; g.C = C; g.ldc = N; g.out_f32 = 1; g.out_scale = out_scale;
	launch_gemm(dtype, g, (hipStream_t)stream);
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}
