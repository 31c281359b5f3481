// Diagnostic (not a test, not product code): cost of a grid-wide barrier + a 64 KB all-to-all exchange between co-resident
// workgroups on all CUs - the floor of one phase boundary in a persistent (one launch per token) decode kernel, to be compared
// with the ~1.5 us kernel boundary + ~2 us first fetch of the multi-launch decode.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tests/diag/grid_barrier.cpp -o tests/diag/grid_barrier.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int SPIN_LIMIT = 1 << 22;      // every wait is bounded: on expiry the kernel flags an error and all later waits fall through

struct Args { unsigned* flags; int bar; unsigned* ctr; unsigned* err; float* buf; float* out; int iters; int mode; int nwg; };

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned* err, unsigned target) {
	__syncthreads();
	bool ok = true;
	if (threadIdx.x == 0) {
		__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		int spins = 0;
		while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
			if (++spins > SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
				__hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				ok = false;
				break;
			}
			__builtin_amdgcn_s_sleep(1);
		}
	}
	__syncthreads();
	return ok;
}

// Flag barrier: no read-modify-write at all.  Each workgroup publishes its own epoch word (sc1 store); wave 0 polls the whole flag
// array (<= 256 words = one 16-byte load per lane) until every word has reached the epoch.  STRIDE spreads the flags over lines.
template <int STRIDE>
__device__ __forceinline__ bool flag_barrier(unsigned* flags, unsigned* err, unsigned epoch, int nwg) {
	__syncthreads();
	bool ok = true;
	if (threadIdx.x < 64) {
		if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x * STRIDE, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		int spins = 0;
		for (;;) {
			bool mine = true;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const int w = threadIdx.x * 4 + j;
				if (w < nwg) mine &= __hip_atomic_load(flags + w * STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
			}
			if (__all(mine)) break;
			if (++spins > SPIN_LIMIT / 4 || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
				__hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				ok = false;
				break;
			}
		}
		__atomic_thread_fence(__ATOMIC_ACQUIRE);   // agent scope by default for device code
	}
	__shared__ int s_ok;
	if (threadIdx.x == 0) s_ok = ok;
	__syncthreads();
	return s_ok;
}

// mode 0: barrier only.  mode 1: + each WG publishes its slice of a 16x1024 f32 panel (plain stores, release/acquire fences) and reads
// the whole panel back after the barrier.  mode 2: same with sc1 (agent-scope relaxed atomic) stores and loads.
__global__ __launch_bounds__(512) void k_bar(Args a) {
	const int tid = threadIdx.x, wg = blockIdx.x;
	const int per = 16384 / a.nwg;
	float acc = 0.f;
	for (int it = 0; it < a.iters; ++it) {
		float* panel = a.buf + (it & 1) * 16384;
		if (a.mode == 1) {
			if (tid < per) panel[wg * per + tid] = acc + it;
		} else if (a.mode == 2) {
			if (tid < per) __hip_atomic_store(panel + wg * per + tid, acc + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		bool ok;
		if (a.bar == 0) ok = grid_barrier(a.ctr, a.err, (unsigned)(it + 1) * a.nwg);
		else if (a.bar == 1) ok = flag_barrier<1>(a.flags, a.err, (unsigned)(it + 1), a.nwg);
		else ok = flag_barrier<16>(a.flags, a.err, (unsigned)(it + 1), a.nwg);
		if (!ok) return;
		if (a.mode == 1) {
			const float4* p4 = (const float4*)panel;
#pragma unroll
			for (int j = 0; j < 8; ++j) { const float4 v = p4[tid + 512 * j]; acc += v.x + v.y + v.z + v.w; }
		} else if (a.mode == 2) {
#pragma unroll
			for (int j = 0; j < 32; ++j) acc += __hip_atomic_load(panel + tid + 512 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		acc *= 1e-6f;
	}
	if (tid == 0) a.out[wg] = acc;
}

int main() {
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("CUs %d\n", cus);
	Args a; CK(hipMalloc(&a.ctr, 256)); CK(hipMalloc(&a.flags, 256 * 64)); CK(hipMalloc(&a.err, 4)); CK(hipMalloc(&a.buf, 2 * 16384 * 4)); CK(hipMalloc(&a.out, 4096));
	hipStream_t s; CK(hipStreamCreate(&s));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const char* names[3] = {"barrier only", "barrier + 64 KB panel exchange (fences)", "barrier + 64 KB panel exchange (sc1 ld/st)"};
	for (int nwg : {64, 128, 256}) {
		if (nwg > cus) continue;
		for (int bar = 0; bar < 3; ++bar)
		for (int mode = 0; mode < 3; ++mode) {
			a.iters = 2000; a.mode = mode; a.nwg = nwg; a.bar = bar;
			float best = 1e9f;
			for (int rep = 0; rep < 3; ++rep) {
				CK(hipMemsetAsync(a.ctr, 0, 256, s)); CK(hipMemsetAsync(a.flags, 0, 256 * 64, s)); CK(hipMemsetAsync(a.err, 0, 4, s)); CK(hipMemsetAsync(a.buf, 0, 2 * 16384 * 4, s));
				CK(hipEventRecord(e0, s));
				k_bar<<<nwg, 512, 0, s>>>(a);
				CK(hipEventRecord(e1, s));
				CK(hipStreamSynchronize(s));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				best = ms < best ? ms : best;
			}
			unsigned err; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
			printf("%3d WGs  %-14s %-44s %7.3f us / phase%s\n", nwg, bar == 0 ? "counter" : bar == 1 ? "flags packed" : "flags 64B", names[mode], best * 1e3f / a.iters, err ? "   (SPIN LIMIT HIT)" : "");
		}
	}
	return 0;
}
