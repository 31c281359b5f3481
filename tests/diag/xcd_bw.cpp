// Diagnostic (not a test, not product code): (a) read bandwidth a SUBSET of the XCDs reaches on cold data -- workgroups are dealt to XCDs
// round-robin by id, so only ids with (id % 8) < nx do work -- and (b) the cost of a barrier among workgroups that all sit on XCD 0
// (atomic counter in that XCD's L2).  Sizing question behind it: would a persistent decode kernel confined to one XCD (exchange through
// its own L2, sub-microsecond barriers) stream a layer's 25 MB of weights fast enough to beat five launches per layer?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tests/diag/xcd_bw.cpp -o tests/diag/xcd_bw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// each working workgroup streams a contiguous share of [base, base + bytes) with 16-byte non-temporal loads, UN in flight per lane
template <int UN>
__global__ __launch_bounds__(256) void k_stream(const u32x4* __restrict__ base, int64_t n16, int nx, unsigned* __restrict__ sink) {
	const int xcd = blockIdx.x & 7;
	if (xcd >= nx) return;
	const int rank = (blockIdx.x >> 3) * nx + xcd, nwork = (gridDim.x >> 3) * nx;
	const int64_t per = (n16 + nwork - 1) / nwork, lo = rank * per, hi = lo + per < n16 ? lo + per : n16;
	unsigned acc = 0;
	for (int64_t i = lo + threadIdx.x; i < hi; i += 256 * UN) {
		u32x4 v[UN];
#pragma unroll
		for (int u = 0; u < UN; ++u) { const int64_t j = i + (int64_t)u * 256; v[u] = j < hi ? __builtin_nontemporal_load(base + j) : u32x4{0, 0, 0, 0}; }
#pragma unroll
		for (int u = 0; u < UN; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
	}
	if (acc == 0x12345678u) sink[0] = acc;
}

constexpr int SPIN_LIMIT = 1 << 20;
// iters barriers among the workgroups on XCD 0 (ids % 8 == 0); others exit.  Bounded spins; err set on expiry.
__global__ __launch_bounds__(256) void k_xcd_barrier(unsigned* ctr, unsigned* err, int iters, int nwork) {
	if ((blockIdx.x & 7) != 0) return;
	for (int it = 1; it <= iters; ++it) {
		__syncthreads();
		if (threadIdx.x == 0) {
			__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			int spins = 0;
			while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it * nwork)) {
				if (++spins > SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
			}
		}
		__syncthreads();
	}
}

int main() {
	const int64_t pool_bytes = (int64_t)1 << 30, chunk = 25 << 20;        // 1 GiB pool, 25 MiB per launch (one GPT-2 layer in bf16)
	u32x4* pool; unsigned *sink, *ctr, *err;
	CK(hipMalloc(&pool, pool_bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&err, 64));
	CK(hipMemset(pool, 1, pool_bytes)); CK(hipMemset(sink, 0, 64));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const int nchunks = (int)(pool_bytes / chunk);
	for (int wg_per_cu : {1, 2, 4}) {
		for (int nx : {1, 2, 4, 8}) {
			const int grid = 8 * 32 * wg_per_cu;          // 32 CUs per XCD
			float best = 1e9f, sum = 0; int n = 0;
			for (int rep = 0; rep < 24; ++rep) {
				const u32x4* p = pool + (int64_t)(rep % nchunks) * (chunk / 16);
				CK(hipEventRecord(a));
				hipLaunchKernelGGL((k_stream<8>), dim3(grid), dim3(256), 0, 0, p, chunk / 16, nx, sink);
				CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
				float ms; CK(hipEventElapsedTime(&ms, a, b));
				if (rep >= 4) { best = ms < best ? ms : best; sum += ms; ++n; }
			}
			printf("stream 25 MiB cold: %d XCD(s) x 32 CUs x %d WG/CU: best %.1f us (%.2f TB/s)  mean %.1f us\n", nx, wg_per_cu, best * 1e3, chunk / (best * 1e-3) / 1e12, sum / n * 1e3);
		}
	}
	for (int nwork : {32, 64}) {
		const int iters = 200;
		for (int rep = 0; rep < 3; ++rep) {
			CK(hipMemset(ctr, 0, 64)); CK(hipMemset(err, 0, 64));
			CK(hipEventRecord(a));
			hipLaunchKernelGGL(k_xcd_barrier, dim3(8 * nwork), dim3(256), 0, 0, ctr, err, iters, nwork);
			CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
			float ms; CK(hipEventElapsedTime(&ms, a, b));
			unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
			if (rep == 2) printf("barrier among %d WGs on XCD 0: %.2f us each (err %u)\n", nwork, ms * 1e3 / iters, e);
		}
	}
	return 0;
}
