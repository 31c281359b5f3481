// Diagnostic (not a test, not product code): can a chain of dependent decode-step-like kernels overlap across TWO queues, with the
// dependency carried by device flags instead of the stream order?  Each kernel stands in for one launch of the AR decode step:
// 256 workgroups x 512 threads; a workgroup streams WBYTES of cold weights (rotating through a 1 GiB buffer), needs ALL 64 KB that
// its predecessor wrote (16 x 1024 f32), and writes its own 256 bytes of the next 64 KB.
//   mode 0  one stream, plain loads / stores: the chain as libttk runs it today (kernel boundary = the dependency)
//   mode 1  kernels alternate between two streams with NO stream edge between them; kernel k+1 issues its weight loads, then polls
//           the arrival counter of kernel k (8 shards, one per blockIdx & 7), then reads the activations with sc1 loads; outputs are
//           sc1 (write-through) stores, drained, then one agent-scope atomic add per workgroup.  Counter k is zeroed by kernel k+3
//           (same queue as its consumer k+1, so that one has completed).  Every spin is bounded and reports through an error word.
//   mode 2  the flag protocol on ONE stream (its cost without any overlap)
// All three as one captured graph of CHAIN kernels.  The payload is checked: kernel k expects every input word to be k and writes
// k + 1, so a stale or early read is counted.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tests/diag/twoq.cpp -o tests/diag/twoq.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NWG = 256, NT = 512, ACT = 16384;            // activations: 16 x 1024 f32
constexpr int SHARDS = 8, SHARD_STRIDE = 32;               // one 128-byte line per shard
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct P {
	const v4u* w; size_t w_off; int w_vec;                // uint4 per thread (WBYTES = w_vec * 16 * NT per workgroup)
	const float* in; float* out;
	unsigned* wait_ctr; unsigned wait_target;               // per shard
	unsigned* done_ctr; unsigned* zero_ctr;
	unsigned* err;                                          // [0] spin timeouts, [1] wrong input words
	float expect; float zero;
};

template <bool FLAG>
__global__ __launch_bounds__(NT) void k_phase(P p) {
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// weights first: they never depend on the predecessor
	v4u wv[8];
#pragma unroll
	for (int j = 0; j < 8; ++j)
		if (j < p.w_vec) wv[j] = __builtin_nontemporal_load(p.w + p.w_off + ((size_t)blockIdx.x * p.w_vec + j) * NT + tid);
		else wv[j] = v4u{0, 0, 0, 0};
	if (FLAG) {
		if (p.zero_ctr && blockIdx.x == 0 && tid < SHARDS) __hip_atomic_store(p.zero_ctr + tid * SHARD_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (p.wait_ctr && wave == 0) {
			unsigned spins = 0;
			for (;;) {
				unsigned v = p.wait_target;
				if (lane < SHARDS) v = __hip_atomic_load(p.wait_ctr + lane * SHARD_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (__all(v >= p.wait_target)) break;
				if (++spins > (1u << 18)) { if (lane == 0) atomicAdd(p.err, 1u); break; }
				__builtin_amdgcn_s_sleep(1);
			}
		}
		__syncthreads();
	}
	// activations: every workgroup reads all of them (8 x 16 B per thread)
	float bad = 0.f, acc = 0.f;
	if (FLAG) {
		__amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, ACT * 4, 0x00020000);
		v4i a[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) a[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + NT * j) * 16, 0, 16);   // aux 16 = sc1
#pragma unroll
		for (int j = 0; j < 8; ++j)
#pragma unroll
			for (int c = 0; c < 4; ++c) { const float f = __int_as_float(a[j][c]); acc += f; bad += (f != p.expect) ? 1.f : 0.f; }
	} else {
		float4 a[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) a[j] = ((const float4*)p.in)[tid + NT * j];
#pragma unroll
		for (int j = 0; j < 8; ++j) { acc += a[j].x + a[j].y + a[j].z + a[j].w; bad += (a[j].x != p.expect) + (a[j].y != p.expect) + (a[j].z != p.expect) + (a[j].w != p.expect); }
	}
	unsigned wsum = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) wsum += wv[j].x ^ wv[j].y ^ wv[j].z ^ wv[j].w;
	if (bad != 0.f) atomicAdd(p.err + 1, 1u);
	const float outv = p.expect + 1.0f + p.zero * (acc + (float)wsum);       // keeps the weight and activation loads alive
	if (tid < 64) {
		if (FLAG) {
			__amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, ACT * 4, 0x00020000);
			__builtin_amdgcn_raw_buffer_store_b32(__float_as_int(outv), ro, (blockIdx.x * 64 + tid) * 4, 0, 16);
		} else {
			p.out[blockIdx.x * 64 + tid] = outv;
		}
	}
	if (FLAG) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (tid == 0 && p.done_ctr) __hip_atomic_fetch_add(p.done_ctr + (blockIdx.x & (SHARDS - 1)) * SHARD_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

int main(int argc, char** argv) {
	const int chain = argc > 1 ? atoi(argv[1]) : 600;
	const int w_vec = argc > 2 ? atoi(argv[2]) : 4;          // 4 -> 32 KB per workgroup, 8 MB per kernel
	const size_t wbytes = (size_t)1 << 30;
	v4u* w; float *a, *b; unsigned *ctr, *err;
	CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
	CK(hipMalloc(&a, ACT * 4)); CK(hipMalloc(&b, ACT * 4));
	const int nctr = chain + 4;
	CK(hipMalloc(&ctr, (size_t)nctr * SHARDS * SHARD_STRIDE * 4)); CK(hipMalloc(&err, 64));
	hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
	hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
	const size_t per_kernel = (size_t)NWG * w_vec * NT;       // uint4 units
	const size_t wrap = wbytes / 16 / per_kernel;
	printf("chain %d, %zu KB of weights per workgroup, %.1f MB per kernel\n", chain, (size_t)w_vec * 16 * NT / 1024, per_kernel * 16 / 1e6);
	for (int mode = 0; mode < 3; ++mode) {
		hipGraph_t g; hipGraphExec_t ge;
		CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
		if (mode == 1) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
		for (int i = 0; i < chain; ++i) {
			P p = {};
			p.w = w; p.w_off = (size_t)(i % wrap) * per_kernel; p.w_vec = w_vec;
			p.in = (i & 1) ? b : a; p.out = (i & 1) ? a : b; p.err = err; p.expect = (float)i; p.zero = 0.f;
			if (mode != 0) {
				p.done_ctr = ctr + (size_t)i * SHARDS * SHARD_STRIDE;
				if (i > 0) { p.wait_ctr = ctr + (size_t)(i - 1) * SHARDS * SHARD_STRIDE; p.wait_target = NWG / SHARDS; }
				if (i >= 3) p.zero_ctr = ctr + (size_t)(i - 3) * SHARDS * SHARD_STRIDE;
			}
			hipStream_t s = (mode == 1 && (i & 1)) ? s1 : s0;
			if (mode == 0) hipLaunchKernelGGL(k_phase<false>, dim3(NWG), dim3(NT), 0, s, p);
			else hipLaunchKernelGGL(k_phase<true>, dim3(NWG), dim3(NT), 0, s, p);
		}
		if (mode == 1) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
		CK(hipStreamEndCapture(s0, &g));
		CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
		float best = 1e9f;
		unsigned herr[2] = {0, 0};
		std::vector<float> h(ACT);
		for (int rep = 0; rep < 6; ++rep) {
			CK(hipMemsetAsync(a, 0, ACT * 4, s0)); CK(hipMemsetAsync(b, 0, ACT * 4, s0));
			CK(hipMemsetAsync(ctr, 0, (size_t)nctr * SHARDS * SHARD_STRIDE * 4, s0)); CK(hipMemsetAsync(err, 0, 64, s0));
			CK(hipStreamSynchronize(s0));
			CK(hipEventRecord(e0, s0)); CK(hipGraphLaunch(ge, s0)); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (rep >= 1 && ms < best) best = ms;
			unsigned e2[2]; CK(hipMemcpy(e2, err, 8, hipMemcpyDeviceToHost)); herr[0] += e2[0]; herr[1] += e2[1];
			CK(hipMemcpy(h.data(), (chain & 1) ? b : a, ACT * 4, hipMemcpyDeviceToHost));
			for (int i = 0; i < ACT; ++i) if (h[i] != (float)chain) { herr[1] += 1; break; }
		}
		const char* names[3] = {"one stream, kernel boundaries", "two streams, device flags", "one stream, device flags"};
		printf("mode %d (%s): %.3f us per kernel   [spin timeouts %u, wrong payload %u]\n", mode, names[mode], best * 1e3 / chain, herr[0], herr[1]);
		CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
	}
	return 0;
}
