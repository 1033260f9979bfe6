// Microbenchmark 2 (development aid): how the cost of a 64 B-per-lane gather depends on how many DISTINCT records the 64
// lanes of a wave ask for (coherent rays share BVH nodes), and on the record size (64 B vs 32 B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int QUADS>   // QUADS x 16 B per record
__global__ __launch_bounds__(256) void k(const float4* __restrict__ table, uint32_t mask, int iters, float* out, int distinct, int adjacent) {
	// `distinct` records per wave; adjacent != 0: lanes sharing a record are neighbours, else interleaved
	const int lane = threadIdx.x & 63;
	const int group = adjacent ? lane / (64 / distinct) : lane % distinct;
	uint32_t s = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + group) * 2654435761u + 12345u;   // same stream within a group
	float acc = 0;
	uint32_t idx = rnd(s) & mask;
	for (int it = 0; it < iters; it++) {
		const float4* p = table + (size_t)idx * QUADS;
		float4 v[QUADS];
#pragma unroll
		for (int q = 0; q < QUADS; q++) v[q] = p[q];
		float sum = 0;
#pragma unroll
		for (int q = 0; q < QUADS; q++) sum += v[q].x + v[q].w;
		acc += sum;
		idx = (rnd(s) ^ (__float_as_uint(v[QUADS - 1].x) & 0)) & mask;
		if (__float_as_uint(v[0].y) == 0xdeadbeef) idx = 0;   // keep the dependence on the loaded data
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
	const int iters = 256, blocks = 256 * 4;
	const uint32_t n = 1u << 17;   // 8 MB of 64 B records
	std::vector<float> h((size_t)n * 16, 1.0f);
	float4* table; float* out;
	CHK(hipMalloc(&table, h.size() * 4)); CHK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	CHK(hipMalloc(&out, blocks * 256 * 4));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	for (int quads : {4, 2}) for (int adjacent : {1, 0}) for (int distinct : {1, 4, 16, 64}) {
		float best = 1e9;
		for (int rep = 0; rep < 3; rep++) {
			CHK(hipEventRecord(e0));
			if (quads == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, table, n - 1, iters, out, distinct, adjacent);
			else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, table, (n * 2) - 1, iters, out, distinct, adjacent);
			CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		printf("record=%2d B  distinct/wave=%2d %s  %.3f ms  %.0f cycles per wave-iteration per CU\n", quads * 16, distinct, adjacent ? "adjacent   " : "interleaved",
		       best, best * 1e-3 * 2.4e9 / ((double)blocks / 256 * 4 * iters));
	}
	return 0;
}
