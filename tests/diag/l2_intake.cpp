// Diagnostic (not a test, not product code): what ONE CU takes in from its XCD's L2, by the way the bytes are requested (round 6: the DDIM GEMMs' k-loops all run at
// 24 KiB (or 48 KiB) per ~765 (1530) shader cycles = ~32 B/clk per CU whatever the instruction order inside a k-step -- is that the LDS-DMA path's rate, or the CU's?).
// Every workgroup (one per CU, 4 or 8 waves) sweeps 24-KiB "tiles" of an L2-resident region, 1 KiB per wave instruction as the GEMM's staging does, `depth` tiles in flight:
//   mode 0  buffer_load_dwordx4 ... lds   (LDS-DMA, what csrc/gemm.hip does)
//   mode 1  buffer_load_dwordx4 to registers, consumed by an xor
//   mode 2  as 1 + ds_write_b128 of every register set (register staging)
//   mode 3  4 pieces by LDS-DMA + 2 to registers per 6 (are the two paths additive?)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tests/diag/l2_intake.cpp -o tests/diag/l2_intake.bin && tests/diag/l2_intake.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ void dma(unsigned voff, __amdgpu_buffer_rsrc_t srd, unsigned soff, unsigned dst) {
	asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void ldr(u32x4& d, unsigned voff, __amdgpu_buffer_rsrc_t srd, unsigned soff) {
	asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// region: `bytes` (power of two) of L2-resident data; tile t of workgroup b starts at ((b * 977 + t) * 24 KiB) % bytes; PIECES pieces per wave per tile
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void k_intake(const char* src, unsigned bytes, int ntiles, unsigned long long* out, unsigned* sink) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr int PIECES = 24 / NW;      // 1-KiB pieces per wave per 24-KiB tile
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
	const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes + 32768u, 0x00020000);
	const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
	const unsigned voff = (unsigned)lane * 16u;
	unsigned acc = 0;
	u32x4 r[3][PIECES];
	auto tile_off = [&](int t) { return (unsigned)(((blockIdx.x * 977u + (unsigned)t) * 24576u) & (bytes - 1)); };
	auto request = [&](int t, int set) {
		const unsigned base = tile_off(t) + (unsigned)wave * PIECES * 1024u;
#pragma unroll
		for (int i = 0; i < PIECES; ++i) {
			const bool to_lds = MODE == 0 || (MODE == 3 && i % 3 != 2);
			if (to_lds) dma(voff, srd, base + i * 1024u, lds0 + (set * 24 + wave * PIECES + i) * 1024u);
			else ldr(r[set][i], voff, srd, base + i * 1024u);
		}
	};
	auto consume = [&](int set) {
#pragma unroll
		for (int i = 0; i < PIECES; ++i) {
			const bool to_lds = MODE == 0 || (MODE == 3 && i % 3 != 2);
			if (to_lds) continue;
			asm volatile("" : "+v"(r[set][i]));      // (orders the use behind the counted wait in front of consume())
			if (MODE == 2) *(u32x4*)(smem + (set * 24 + wave * PIECES + i) * 1024 + lane * 16) = r[set][i];
			else acc ^= r[set][i][0] ^ r[set][i][1] ^ r[set][i][2] ^ r[set][i][3];
		}
	};
	__syncthreads();
	const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
	request(0, 0); request(1, 1);
	for (int t = 0; t < ntiles; t += 3) {      // two tiles in flight behind the one being consumed
		request(t + 2, 2); wait_vm<2 * PIECES>(); consume(0);
		request(t + 3, 0); wait_vm<2 * PIECES>(); consume(1);
		request(t + 4, 1); wait_vm<2 * PIECES>(); consume(2);
	}
	wait_vm<0>();
	const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
	if (MODE == 2 || MODE == 0 || MODE == 3) { __syncthreads(); acc ^= *(unsigned*)(smem + lane * 4); }
	if (acc == 0x12345678u) sink[0] = acc;
	if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = t1 - t0; }
}

template <int MODE, int NW>
static int run(const char* name, const char* src, unsigned bytes, int ntiles, int nwg, unsigned long long* out, unsigned* sink) {
	CK(hipFuncSetAttribute((const void*)k_intake<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
	for (int rep = 0; rep < 3; ++rep) { k_intake<MODE, NW><<<nwg, 64 * NW, 72 * 1024>>>(src, bytes, ntiles, out, sink); }
	CK(hipDeviceSynchronize());
	std::vector<unsigned long long> h(2 * nwg);
	CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
	std::vector<double> bpc, gbs;
	for (int b = 0; b < nwg; ++b) { const double by = (double)(ntiles + 2) * 24576.0; bpc.push_back(by / (double)h[2 * b]); gbs.push_back(by / ((double)h[2 * b + 1] * 10.0)); }
	std::sort(bpc.begin(), bpc.end()); std::sort(gbs.begin(), gbs.end());
	printf("%-46s %d waves, region %5u KiB: median %.1f B/clk per CU (p10 %.1f, p90 %.1f), %.1f GB/s per CU\n", name, NW, bytes >> 10, bpc[nwg / 2], bpc[nwg / 10], bpc[nwg * 9 / 10], gbs[nwg / 2]);
	return 0;
}

int main(int argc, char** argv) {
	const int ntiles = 600, nwg = argc > 1 ? atoi(argv[1]) : 256;
	char* src; unsigned long long* out; unsigned* sink;
	CK(hipMalloc(&src, 64u << 20)); CK(hipMemset(src, 1, 64u << 20)); CK(hipMalloc(&out, 2 * 1024 * 8)); CK(hipMalloc(&sink, 4));
	for (unsigned bytes : {1u << 20, 2u << 20, 16u << 20}) {      // 1-2 MiB: every XCD's L2 holds it; 16 MiB: beyond one L2 (4 MiB), inside the Infinity Cache
		if (run<0, 4>("LDS-DMA (buffer_load ... lds)", src, bytes, ntiles, nwg, out, sink)) return 1;
		if (run<1, 4>("buffer_load to registers", src, bytes, ntiles, nwg, out, sink)) return 1;
		if (run<2, 4>("buffer_load to registers + ds_write_b128", src, bytes, ntiles, nwg, out, sink)) return 1;
		if (run<3, 4>("4 of 6 pieces LDS-DMA + 2 to registers", src, bytes, ntiles, nwg, out, sink)) return 1;
		if (run<0, 8>("LDS-DMA (buffer_load ... lds)", src, bytes, ntiles, nwg, out, sink)) return 1;
		if (run<1, 8>("buffer_load to registers", src, bytes, ntiles, nwg, out, sink)) return 1;
		if (run<2, 8>("buffer_load to registers + ds_write_b128", src, bytes, ntiles, nwg, out, sink)) return 1;
	}
	return 0;
}
