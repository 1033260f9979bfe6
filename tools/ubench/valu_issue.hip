// tools/ubench/valu_issue.hip -- what one SIMD of an MI355X issues per cycle for the instruction kinds the packet walks are made of,
// at 1..8 waves per SIMD: the denominator of bench.py's valu_issue roofline.  Every wave runs ITER x 64 independent instructions of one
// kind and stamps s_memtime around them; cycles per wave-instruction per SIMD = elapsed / (instructions x waves on the SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 2000
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float a, float b) {
	float r[16];
	v2f p[8];
	for (int i = 0; i < 16; i++) r[i] = a + i + threadIdx.x;
	for (int i = 0; i < 8; i++) p[i] = v2f{ a + i, b - i };
	unsigned long long m = 0;
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < ITER; it++) {
#pragma unroll
		for (int u = 0; u < 4; u++) {
#pragma unroll
			for (int i = 0; i < 16; i++) {
				if (KIND == 0) r[i] = __builtin_fmaf(r[i], a, b);
				if (KIND == 1) r[i] = __builtin_fminf(r[i], b + i);
				if (KIND == 2) { p[i & 7] = p[i & 7] * v2f{ a, a }; }
				if (KIND == 3) r[i] = r[i] * a;
				if (KIND == 4) { m += __ballot(r[i] > b + it); r[i] += 1.0f; }
				if (KIND == 5) r[i] = (float)((double)r[i] * (double)a);
			}
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	float s = (float)m;
	for (int i = 0; i < 16; i++) s += r[i];
	for (int i = 0; i < 8; i++) s += p[i].x + p[i].y;
	out[blockIdx.x * 256 + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}
template <int KIND> void run(const char* name, int instrPerInner) {
	int cus = 256;
	hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
	for (int perCU : { 1, 2, 3, 4, 8 }) {
		const int blocks = cus * perCU;
		float* out; unsigned long long* cyc;
		hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		k<KIND><<<blocks, 256>>>(out, cyc, 1.0001f, 0.5f);
		hipEventRecord(e0);
		k<KIND><<<blocks, 256>>>(out, cyc, 1.0001f, 0.5f);
		hipEventRecord(e1); hipDeviceSynchronize();
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		std::vector<unsigned long long> h((size_t)blocks * 4);
		hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
		double mean = 0; for (auto c : h) mean += (double)c; mean /= (double)h.size();
		const double instr = (double)ITER * 64 * instrPerInner / 1.0;   // per wave (64 = 4 x 16 unrolled)
		// one workgroup = 4 waves = one per SIMD, so waves per SIMD = perCU
		printf("%-22s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (memtime ticks: %.0f per wave), kernel %.3f ms -> %.1f G wave-instr/s chip\n",
		       name, perCU, mean / (instr * perCU), mean, ms, instr * blocks * 4 / (ms * 1e-3) / 1e9);
		hipFree(out); hipFree(cyc);
	}
}
int main() {
	run<0>("v_fma_f32", 1);
	run<1>("v_min_f32", 1);
	run<2>("v_pk_mul_f32", 1);
	run<3>("v_mul_f32", 1);
	run<4>("v_cmp+s_add (ballot)", 2);
	run<5>("v_cvt/v_mul_f64/v_cvt", 3);
	return 0;
}
