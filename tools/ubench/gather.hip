// Microbenchmark (development aid, not part of the library): cost of fetching one 64 B record per lane from a table
// with random per-lane indices, the access pattern of a divergent BVH walk.
//   A: every lane issues 4 x 16 B loads for its own record                       (what the traversal did)
//   B: the 4 lanes of a quad fetch one record together (16 B each), 4 rounds, parts moved with quad DPP
//   C: table staged in LDS, every lane reads its record with 4 x ds_read_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

__global__ __launch_bounds__(256) void kmask(const float4* __restrict__ table, uint32_t mask, int iters, float* out, int activeOf64) {
	uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
	float acc = 0;
	uint32_t idx = rnd(s) & mask;
	// lanes are spread so that active lanes are not adjacent (stride pattern)
	const int lane = threadIdx.x & 63;
	const bool on = ((lane * activeOf64) & 63) < activeOf64 || activeOf64 == 64;
	if (on) {
		for (int it = 0; it < iters; it++) {
			const float4* p = table + (size_t)idx * 4;
			float4 a = p[0], b = p[1], c = p[2], d = p[3];
			acc += a.x + b.y + c.z + d.w;
			idx = (rnd(s) ^ __float_as_uint(d.x)) & mask;
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ table, uint32_t mask, int iters, float* out) {
	extern __shared__ float4 lds[];
	uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
	if (MODE == 2) {
		for (uint32_t i = threadIdx.x; i < (mask + 1) * 4; i += 256) lds[i] = table[i];
		__syncthreads();
	}
	float acc = 0;
	uint32_t idx = rnd(s) & mask;
	for (int it = 0; it < iters; it++) {
		float4 a, b, c, d;
		if (MODE == 0) {
			const float4* p = table + (size_t)idx * 4;
			a = p[0]; b = p[1]; c = p[2]; d = p[3];
		} else if (MODE == 1) {
			const int q = threadIdx.x & 3;
			float4 part[4];
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const uint32_t want = __shfl(idx, (threadIdx.x & 60) + r, 64);   // the quad's lane r index
				part[r] = table[(size_t)want * 4 + q];
			}
			// lane q needs part q of... every lane j of the quad holds chunk j of record r in part[r]; lane r wants chunks 0..3 of record r
			float4 got[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {   // chunk j comes from lane j of the quad, its part[my r]
				float4 v;
				const int r = threadIdx.x & 3;
				// select part[r] without dynamic register indexing
				float4 sel = r == 0 ? part[0] : r == 1 ? part[1] : r == 2 ? part[2] : part[3];
				(void)sel;
				// what lane j holds for MY record is its part[r_me]; fetch it by shuffling each candidate and selecting
				float4 c0 = part[0], c1 = part[1], c2 = part[2], c3 = part[3];
				const int src = (threadIdx.x & 60) + j;
				float4 s0, s1, s2, s3;
				s0.x = __shfl(c0.x, src, 64); s0.y = __shfl(c0.y, src, 64); s0.z = __shfl(c0.z, src, 64); s0.w = __shfl(c0.w, src, 64);
				s1.x = __shfl(c1.x, src, 64); s1.y = __shfl(c1.y, src, 64); s1.z = __shfl(c1.z, src, 64); s1.w = __shfl(c1.w, src, 64);
				s2.x = __shfl(c2.x, src, 64); s2.y = __shfl(c2.y, src, 64); s2.z = __shfl(c2.z, src, 64); s2.w = __shfl(c2.w, src, 64);
				s3.x = __shfl(c3.x, src, 64); s3.y = __shfl(c3.y, src, 64); s3.z = __shfl(c3.z, src, 64); s3.w = __shfl(c3.w, src, 64);
				v = r == 0 ? s0 : r == 1 ? s1 : r == 2 ? s2 : s3;
				got[j] = v;
			}
			a = got[0]; b = got[1]; c = got[2]; d = got[3];
		} else {
			const float4* p = lds + (size_t)idx * 4;
			a = p[0]; b = p[1]; c = p[2]; d = p[3];
		}
		acc += a.x + b.y + c.z + d.w;
		idx = (rnd(s) ^ __float_as_uint(d.x)) & mask;   // dependent on the loaded record, like a tree walk
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
	const int iters = 256, blocks = 256 * 6;
	for (int logN = 8; logN <= 17; logN += 3) {
		const uint32_t n = 1u << logN;
		std::vector<float> h((size_t)n * 16);
		for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 97) * 0.01f;
		float4* table; float* out;
		CHK(hipMalloc(&table, h.size() * 4)); CHK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
		CHK(hipMalloc(&out, blocks * 256 * 4));
		hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
		for (int mode = 0; mode < 3; mode++) {
			if (mode == 2 && n * 64 > 64 * 1024) continue;
			float best = 1e9;
			for (int rep = 0; rep < 4; rep++) {
				CHK(hipEventRecord(e0));
				size_t shm = mode == 2 ? (size_t)n * 64 : 0;
				if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), shm, 0, table, n - 1, iters, out);
				if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), shm, 0, table, n - 1, iters, out);
				if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), shm, 0, table, n - 1, iters, out);
				CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
				float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			double fetches = (double)blocks * 256 * iters;
			printf("records=%7u (%6.0f KB) mode=%d  %.3f ms  %.1f G record-fetches/s\n", n, n * 64 / 1024.0, mode, best, fetches / best / 1e6);
		}
		for (int act : {64, 32, 16, 8}) for (int bl : {256, 256 * 2, 256 * 6}) {
			float best = 1e9;
			for (int rep = 0; rep < 3; rep++) {
				CHK(hipEventRecord(e0));
				hipLaunchKernelGGL(kmask, dim3(bl), dim3(256), 0, 0, table, n - 1, iters, out, act);
				CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
				float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			printf("   active %2d/64 blocks/CU=%d  %.3f ms  -> %.1f cycles per wave-iteration per CU (2.4GHz)\n", act, bl / 256, best, best * 1e-3 * 2.4e9 / ((double)bl / 256 * 4 * iters));
		}
		CHK(hipFree(table)); CHK(hipFree(out));
	}
	return 0;
}
