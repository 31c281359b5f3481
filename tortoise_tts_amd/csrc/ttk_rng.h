// The Exp(1) noise of torch.multinomial, reproduced bit for bit outside torch.
//
// `q.exponential_(1)` on a CUDA/HIP tensor of `numel` floats (ATen native/cuda/DistributionTemplates.h, distribution_nullary_kernel +
// transformation::exponential): a grid of `threads` = 256 * min(SMs * maxThreadsPerSM / 256, ceil(numel / 256)) threads, thread idx initialises
// a Philox4_32_10 state with (seed, subsequence = idx, offset) and draws ONE float4 per round `it`; element li of the flattened tensor takes word
// (li % (4 * threads)) / threads of the draw of thread li % threads in round li / (4 * threads).  On ROCm the state and the uniform conversion
// are rocRAND's (hiprand_uniform4 -> rocrand_uniform4: 2^-32 + x * 2^-32, in (0, 1]); the exponential is -log(u) with log(1) replaced by
// -eps/2, `log` being torch_device_log below.  The generator offset advances by 4 * rounds per call.  The device functions below ARE rocRAND's header
// implementations, so only the indexing is restated here; tortoise_tts_amd/autoregressive.py compares this against torch on the device before
// it uses it and keeps drawing with torch.exponential_ otherwise.
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>
#include <rocrand/rocrand_uniform.h>
#include <stdint.h>

namespace ttk {

struct RngArgs {            // device-resident, written by the host at the start of a generation (all int64: one small torch tensor)
	int64_t seed;           // torch generator seed
	int64_t offset0;        // generator offset before the first draw of the generation
	int64_t threads;        // 256 * grid of torch's launch for this numel
	int64_t step;           // offset advance per draw: 4 * rounds
	int64_t row0;           // first row of this rank's candidates inside the [C, V] tensor torch would fill (candidate shards)
	int64_t group;          // > 0: the batch holds several text lines of `group` candidates each, every line drawing the SAME [group, V] noise
	                        // (the reference reseeds to 0 for every line): row m uses noise row m mod group
};

// at::log of the torch wheel on this image (ATen/NumericUtils.h; __HIP_ARCH__ is not a HIP macro, so its `::log(x)` branch is the one
// compiled): OCML's log_f32.  NOT this compiler's logf / __logf -- clang 22 expands those inline to a more accurate sequence that differs
// from torch in the last bit for a third of the arguments -- and not the hardware log2 times ln 2 either (3 % differ).  Measured on
// gfx950 over 131104 values: tests/diag/rng_probe.py, tests/diag/rng_log.cpp; pinned by tests/test_gpu_rng.py.
extern "C" __device__ float __ocml_log_f32(float);
__device__ __forceinline__ float torch_device_log(float u) { return __ocml_log_f32(u); }

// element `li` of the draw number `draw` (0-based within the generation)
__device__ __forceinline__ float torch_exponential_at(const RngArgs& a, int64_t draw, int64_t li) {
	const int64_t per_round = 4 * a.threads;
	const int64_t it = li / per_round, r = li - it * per_round;
	const int ii = (int)(r / a.threads);
	const int64_t idx = r - (int64_t)ii * a.threads;
	rocrand_state_philox4x32_10 st;
	rocrand_init((unsigned long long)a.seed, (unsigned long long)idx, (unsigned long long)(a.offset0 + draw * a.step + 4 * it), &st);
	const float4 u4 = rocrand_uniform4(&st);
	const float u = ii == 0 ? u4.x : (ii == 1 ? u4.y : (ii == 2 ? u4.z : u4.w));
	const float lg = u >= 1.0f - FLT_EPSILON / 2 ? -FLT_EPSILON / 2 : torch_device_log(u);
	return -1.0f / 1.0f * lg;
}

}  // namespace ttk
