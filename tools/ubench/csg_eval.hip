// tools/ubench/csg_eval.hip -- what one evaluation of a CSG program (csgEval, rt_device.h) costs a wave that is alone on its SIMD, and
// what the chip evaluates per second with every SIMD full: the C3 cell (difference(intersection(box, sphere), union(box, union(box,
// box))): 9 nodes), its five leaves without the operators, a single box, a single sphere.  Rays: a 256 x 256 grid of directions from
// a camera above and in front of the cell, so that lists of 0 .. 3 intervals all occur.  Prints a checksum of (t, tag) per program:
// a change to csgEval that changes it has changed results.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../include -I../../raytracer_amd/csrc -o csg_eval csg_eval.hip && ./csg_eval
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "rt_scene.h"
#include "rt_kernels.h"
#include "rt_device.h"
using namespace rtd;

#define EVALS 8   // evaluations per lane in the timed loop
__global__ __launch_bounds__(256) void k(DevScene S, int prog, int side, float* tOut, uint32_t* tagOut, unsigned long long* cyc) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	const int px = i % side, py = (i / side) % side;
	const f3 e = mk3(0.3f, 1.1f, 2.2f);
	unsigned long long h = 0;
	float tsum = 0;
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int r = 0; r < EVALS; r++) {
		const f3 d = mk3(-1.3f + 2.0f * (px + 0.37f * r) / side, -2.65f + 2.0f * (py + 0.21f * r) / side, -6.2f);
		uint32_t tag = 0;
		const float t = csgEval(S, S.csgs[prog], e, d, (r & 1) != 0, tag);
		tsum += t; h = h * 1000003ull + __float_as_uint(t) * 31ull + tag;
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	tOut[i] = tsum; tagOut[i] = (uint32_t)(h ^ (h >> 32));
	if ((threadIdx.x & 63) == 0) cyc[i >> 6] = t1 - t0;
}

int main() {
	const float s = 0.45f, cx = 0, cz = -4.0f;
	auto box = [&](float ax, float ay, float az, float bx, float by, float bz) {
		rt_csg_node n{}; n.kind = RT_CSG_BOX; n.mesh = -1;
		n.f[0] = cx + ax * s; n.f[1] = cx + bx * s; n.f[2] = -1 + ay * s; n.f[3] = -1 + by * s; n.f[4] = cz + az * s; n.f[5] = cz + bz * s;
		return n;
	};
	auto sphere = [&]() { rt_csg_node n{}; n.kind = RT_CSG_SPHERE; n.mesh = -1; n.f[0] = cx; n.f[1] = -1 + s; n.f[2] = cz; n.f[3] = 1.35f * s; return n; };
	auto op = [&](int kind) { rt_csg_node n{}; n.kind = kind; n.mesh = -1; return n; };
	const rt_csg_node A = box(-1, 0, -1, 1, 2, 1), B1 = box(-0.5f, 0.5f, -1.5f, 0.5f, 1.5f, 1.5f), B2 = box(-1.5f, 0.5f, -0.5f, 1.5f, 1.5f, 0.5f), B3 = box(-0.5f, -0.5f, -0.5f, 0.5f, 2.5f, 0.5f);
	std::vector<rt_csg_node> nodes;
	std::vector<rt_csg> csgs;
	auto program = [&](std::vector<rt_csg_node> p) { rt_csg c{}; c.node_first = (int)nodes.size(); c.node_count = (int)p.size(); csgs.push_back(c); nodes.insert(nodes.end(), p.begin(), p.end()); };
	program({ A, sphere(), op(RT_CSG_INTERSECTION), B1, B2, B3, op(RT_CSG_UNION), op(RT_CSG_UNION), op(RT_CSG_DIFFERENCE) });
	program({ A, sphere(), B1, B2, B3 });
	program({ A });
	program({ sphere() });
	program({ A, sphere(), op(RT_CSG_INTERSECTION) });
	program({ B1, B2, B3, op(RT_CSG_UNION), op(RT_CSG_UNION) });
	const char* names[] = { "C3 cell (9 nodes)", "its 5 leaves", "one box", "one sphere", "box ^ sphere", "union of 3 boxes" };
	DevScene S{};
	rt_csg* dc; rt_csg_node* dn;
	hipMalloc(&dc, csgs.size() * sizeof(rt_csg)); hipMalloc(&dn, nodes.size() * sizeof(rt_csg_node));
	hipMemcpy(dc, csgs.data(), csgs.size() * sizeof(rt_csg), hipMemcpyHostToDevice);
	hipMemcpy(dn, nodes.data(), nodes.size() * sizeof(rt_csg_node), hipMemcpyHostToDevice);
	S.csgs = dc; S.csg_nodes = dn; S.csg_cap = RTD_CSG_CAP; S.csg_nodes_max = 9;
	const int side = 256, n = side * side;
	float* tOut; uint32_t* tagOut; unsigned long long* cyc;
	hipMalloc(&tOut, (size_t)n * 64 * 4); hipMalloc(&tagOut, (size_t)n * 64 * 4); hipMalloc(&cyc, (size_t)n * 8);
	int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
	for (int p = 0; p < (int)csgs.size(); p++) {
		// (a) one wave per SIMD on a quarter of the chip: latency of a lone wave
		k<<<1, 64>>>(S, p, side, tOut, tagOut, cyc);
		hipDeviceSynchronize();
		k<<<1, 64>>>(S, p, side, tOut, tagOut, cyc);
		hipDeviceSynchronize();
		unsigned long long lone = 0; hipMemcpy(&lone, cyc, 8, hipMemcpyDeviceToHost);
		// (b) the whole grid: throughput, checksum
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		k<<<n / 256, 256>>>(S, p, side, tOut, tagOut, cyc);
		hipEventRecord(e0);
		const int reps = 20;
		for (int r = 0; r < reps; r++) k<<<n / 256, 256>>>(S, p, side, tOut, tagOut, cyc);
		hipEventRecord(e1); hipDeviceSynchronize();
		float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
		std::vector<uint32_t> tags(n); std::vector<float> ts(n);
		hipMemcpy(tags.data(), tagOut, (size_t)n * 4, hipMemcpyDeviceToHost); hipMemcpy(ts.data(), tOut, (size_t)n * 4, hipMemcpyDeviceToHost);
		unsigned long long sum = 0; int hits = 0;
		for (int i = 0; i < n; i++) { uint32_t b; memcpy(&b, &ts[i], 4); sum = sum * 1000003ull + tags[i] + b; hits += ts[i] != 0; }
		printf("%-20s lone wave: %7.0f cycles per evaluation;  65 536 lanes x %d: %.3f ms -> %.2f G evaluations/s;  lanes with a hit %d, checksum %016llx\n",
		       names[p], (double)lone / EVALS, EVALS, ms, (double)n * EVALS / (ms * 1e-3) / 1e9, hits, sum);
	}
	return 0;
}
