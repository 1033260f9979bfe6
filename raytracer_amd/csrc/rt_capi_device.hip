// rt_capi_device.hip -- device half of the C-ABI (include/rt_capi.h): scene upload and the host loop that drives
// the wavefront pipeline of rt_kernels.hip.  Compiled by hipcc; no CPU fallback exists: without a usable HIP
// device every entry point here fails with RT_EHIP.
#include "rt_capi.h"
#include "rt_kernels.h"
#include "host/host_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <string>
#include <utility>
#include <vector>

namespace {

#define RT_MAX_LANES 8
#define RT_MAX_LEVELS 32   /* max_depth <= 30 is checked by the entry points: levels 0 .. 30 */
#define RT_MAX_INFLIGHT 8
#define HIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { rt::set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); return RT_EHIP; } } while (0)

// launch errors since the last look, with the kernel's name (rt_kernels.hip: noteLaunch), then anything else the runtime holds
int launchesOk() {
	if (const char* e = rtk_take_launch_error()) { rt::set_error(e); return RT_EHIP; }
	HIPCHK(hipGetLastError());
	return RT_OK;
}

uint32_t envU32(const char* name, uint32_t dflt) {
	const char* v = getenv(name);
	if (!v || !*v) return dflt;
	char* end = nullptr;
	unsigned long long x = strtoull(v, &end, 10);
	return end == v || x > 0x7fffffffull ? dflt : (uint32_t)x;   // not a number: the default (0 is a value: RT_PACKET_LEVELS=0)
}

}  // namespace

// ---- rt_intersect_batch_static: the entry point that allocates nothing (include/rt_capi.h) ------------------------------
// Its scratch -- the ray queue, the hit queue and the counters that rt_intersect_batch takes from the handle's lane 0 -- is
// this file-scope device array: every GPU the library runs on holds its own copy, at its own address, until the process
// ends.  In host code the array's NAME is only a stand-in; the device address comes from hipGetSymbolAddress, once per
// device (staticScratchFor, called by rt_scene_upload with that device current), and only that pointer is ever handed to
// kernels and memsets.
struct StaticScratch {
	float ray[RQ_COMPS * RT_STATIC_MAX_RAYS];
	float hit[HQ_COMPS * RT_STATIC_MAX_RAYS];
	RtCounters ctr;
};
static __device__ __align__(256) unsigned char scratch[sizeof(StaticScratch)];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratchMtx;
StaticScratch* g_scratchPtr[kMaxDevices] = {};
// the address of `scratch` on `device`, which must be the current device
int staticScratchFor(int device, StaticScratch** out) {
	*out = nullptr;
	if (device < 0 || device >= kMaxDevices) { rt::set_error("device index beyond the static scratch table"); return RT_EINVAL; }
	std::lock_guard<std::mutex> g(g_scratchMtx);
	if (!g_scratchPtr[device]) {
		void* p = nullptr;
		HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
		if (!p) { rt::set_error("hipGetSymbolAddress returned a null address for the static scratch"); return RT_EHIP; }
		g_scratchPtr[device] = static_cast<StaticScratch*>(p);
	}
	*out = g_scratchPtr[device];
	return RT_OK;
}
}  // namespace

struct rt_scene {
	int device = 0;
	StaticScratch* staticScratch = nullptr;   // this device's copy of `scratch` (rt_intersect_batch_static)
	bool staticWarm = false;                  // rt_intersect_batch_static has run once on this handle outside a stream capture (its kernel variants loaded, their occupancy cached); under mtx
	DevScene dev{};
	bool hasCsg = false;
	int nCsg = 0;                   // CSG objects in the scene (rtk_shadow_area: few enough to defer their evaluation)
	bool hasTransmissive = false;   // some material transmits: k_shade queues requests for k_transmit
	bool hasReflective = false;     // some material reflects (with hasTransmissive: how many children a hit can have)
	// some textured triangle reflects: its hits leave records for k_texel_fix (RtTexFix in rt_kernels.h).  The per-pixel list heads
	// are shared by the lanes like the accumulators (shards own disjoint pixels), the records are per lane.
	bool texelFix = false;
	uint32_t* texHeads = nullptr;
	uint32_t texCap = 0;
	std::vector<void*> allocs;
	// lights, split by how their shadow rays are generated
	int32_t* pointLights = nullptr;
	uint32_t nPointLights = 0;
	std::vector<std::pair<int, uint32_t>> areaLights;   // (light index, samples)
	uint64_t shadowRaysPerSurface = 0;
	// A lane = everything one in-flight pixel batch needs: its queues, counters, events and (lanes > 0) its own stream.
	// A frame is rendered as `lanes.size()` interleaved tile shards at once, one per lane, each driven by its own
	// host thread: the kernels of a level have long tails (a few packets walk ten times the average), and with several
	// independent shards in flight the tail of one kernel is filled by the kernels of the others.
	struct Lane {
		RtQueue ray[2]{}, hitq{}, surfq{}, transq{};
		RtCounters* ctr = nullptr;       // device
		RtCounters* ctrHost = nullptr;   // pinned
		// One counter block per level of the ray tree, for the level loop that runs ahead of the host (Run::levelsAhead): the
		// kernels of level L append into block L and read level L - 1's count from block L - 1, so no block is reset or reused
		// while kernels queued behind it still need it; the host reads block L back (pinned mirror, event) when it gets there.
		RtCounters* lvl = nullptr;       // device, RT_MAX_LEVELS blocks
		RtLevelReport* report = nullptr;      // pinned, mapped: one slot per level, written by k_publish
		RtLevelReport* reportDev = nullptr;   // the same memory as the device addresses it
		uint32_t reportSeq = 0;               // last sequence number handed out (a slot is fresh when it shows the number it was queued with)
		hipEvent_t ev[6]{};              // 0,1 whole call; 2,3 nearest kernel; 4,5 shadow kernels
		hipStream_t stream = nullptr;    // owned, non-blocking (lane 0 also runs on the caller's stream)
		RtTexRec* texRec = nullptr;      // texelFix only
	};
	RtTexFix texFix(const Lane& ln) const { return RtTexFix{ texelFix ? ln.texRec : nullptr, texHeads, texCap, &ln.ctr->texrec_count }; }
	uint32_t qcap = 0;               // entries per queue (rays, hits, surfaces, requests of one lane's batch)
	uint32_t qcapMax = 0;            // ... and how far rt_render_submit may grow them for large frames
	int maxBlocks = 2048;            // grid cap of the persistent kernels
	bool transmitAnalytic = false;   // the scene has no CSG object and no triangle of it transmits: exit searches are one analytic test per internal reflection (k_level does them inline)
	RtSmallScene small;              // the scene's walk / shading arrays as one blob for k_level's LDS copy (small scenes only)
	int maxBlocksPacketNear = 2048;  // ... of the nearest-hit packet kernel (see rt_scene_upload)
	int maxBlocksPacketShadow = 2048;   // ... and of the shadow packet kernels (one light per packet / two)
	int maxBlocksPacketShadow2 = 2048;
	std::vector<Lane> lanes;
	RtAccum accum{};
	// RT_FLAG_WHOLE_FRAME: a frame rendered on one lane owns no fixed set of pixels, so frames on different lanes may not share
	// accumulators as shards do -- every lane has its own for such frames (24 B per pixel, allocated on first use)
	RtAccum laneAccum[RT_MAX_LANES]{};
	int nextWholeLane = 0;
	// Frames are rendered asynchronously: rt_render_submit queues one shard of the frame on every lane (shard k = the
	// tiles t of the call with t % (tile_ranks * lanes) == tile_rank + tile_ranks * k), each lane's worker thread drives
	// its shards in submission order, and rt_render_wait joins them.  Lanes never wait for each other, so the tail of
	// one shard's kernels overlaps the next shard -- of the same frame or, with several frames in flight, of the next.
	struct Job {
		uint64_t id = 0;                 // 0 = slot free
		rt_camera cam{};
		int W = 0, H = 0, aaX = 1, aaY = 1, maxDepth = 0, tileRank = 0, tileRanks = 1, nShards = 1;
		bool whole = false; int lane = 0;   // RT_FLAG_WHOLE_FRAME: the one lane that renders the frame (results in slot 0 of the per-shard arrays)
		uint32_t seed = 0, flags = 0;
		float* rgb = nullptr;
		hipStream_t st = nullptr;
		hipEvent_t evStart = nullptr, evDone[RT_MAX_LANES]{};
		int remaining = 0;               // lanes that have not finished their part yet
		int rc[RT_MAX_LANES]{};
		std::string err[RT_MAX_LANES];
		rt_stats part[RT_MAX_LANES]{};
	};
	Job jobs[RT_MAX_INFLIGHT];
	uint64_t nextId = 1;
	std::deque<Job*> laneQueue[RT_MAX_LANES];
	std::vector<std::thread> workers;
	std::mutex mtx;
	std::condition_variable cvWork, cvDone;
	bool stopping = false;
	hipEvent_t lane0Last = nullptr;      // end of lane 0's previous shard and the stream it ran on (lane 0's worker only)
	hipStream_t lane0LastStream = nullptr;

	template <class T> int upload(const T* src, int64_t count, const T** dst) {
		*dst = nullptr;
		size_t bytes = (size_t)std::max<int64_t>(count, 1) * sizeof(T);
		void* p = nullptr;
		HIPCHK(hipMalloc(&p, bytes));
		allocs.push_back(p);
		if (count > 0) HIPCHK(hipMemcpy(p, src, (size_t)count * sizeof(T), hipMemcpyHostToDevice));
		else HIPCHK(hipMemset(p, 0, bytes));
		*dst = static_cast<const T*>(p);
		return RT_OK;
	}
	int allocQueue(RtQueue& q, int comps) {
		void* p = nullptr;
		HIPCHK(hipMalloc(&p, (size_t)comps * qcap * sizeof(float)));
		allocs.push_back(p);
		q.f = static_cast<float*>(p); q.cap = qcap;
		return RT_OK;
	}
	int allocLaneQueues(Lane& ln) {
		if (int rc = allocQueue(ln.ray[0], RQ_COMPS)) return rc;
		if (int rc = allocQueue(ln.ray[1], RQ_COMPS)) return rc;
		if (int rc = allocQueue(ln.hitq, HQ_COMPS)) return rc;
		if (int rc = allocQueue(ln.surfq, SQ_COMPS)) return rc;
		if (hasTransmissive) { if (int rc = allocQueue(ln.transq, TQ_COMPS)) return rc; }
		return RT_OK;
	}
	// Larger queues for every lane (a frame whose share per lane does not fit the present ones in a batch or two: see rt_render_submit).
	// The lanes must be idle and the GPU through with their work (drainLocked): the old buffers are released here.
	int freeLaneQueues() {
		for (Lane& ln : lanes)
			for (RtQueue* q : { &ln.ray[0], &ln.ray[1], &ln.hitq, &ln.surfq, &ln.transq })
				if (q->f) {
					allocs.erase(std::remove(allocs.begin(), allocs.end(), (void*)q->f), allocs.end());
					HIPCHK(hipFree(q->f));
					*q = RtQueue{};
				}
		return RT_OK;
	}
	// (if the device cannot give that much -- other handles, other processes -- the size is halved until it can: at worst the old size,
	// whose memory was just released, and the frame runs in more batches)
	int growQueues(uint32_t cap) {
		const uint32_t old = qcap;
		for (;;) {
			if (int rc = freeLaneQueues()) return rc;
			qcap = cap;
			int rc = RT_OK;
			for (Lane& ln : lanes) if ((rc = allocLaneQueues(ln))) break;
			if (rc == RT_OK) return RT_OK;
			(void)hipGetLastError();
			if (cap <= old) return rc;   // not even the size it had
			cap = std::max(old, cap / 2);
		}
	}
	int initLane(Lane& ln, bool ownStream) {
		if (int rc = allocLaneQueues(ln)) return rc;
		void* p = nullptr;
		if (texelFix) {
			HIPCHK(hipMalloc(&p, (size_t)texCap * sizeof(RtTexRec)));
			allocs.push_back(p);
			ln.texRec = static_cast<RtTexRec*>(p);
		}
		HIPCHK(hipMalloc(&p, sizeof(RtCounters)));
		allocs.push_back(p);
		ln.ctr = static_cast<RtCounters*>(p);
		HIPCHK(hipHostMalloc((void**)&ln.ctrHost, sizeof(RtCounters), hipHostMallocDefault));
		HIPCHK(hipMalloc(&p, sizeof(RtCounters) * RT_MAX_LEVELS));
		allocs.push_back(p);
		ln.lvl = static_cast<RtCounters*>(p);
		HIPCHK(hipHostMalloc((void**)&ln.report, sizeof(RtLevelReport) * RT_MAX_LEVELS, hipHostMallocMapped));
		memset(ln.report, 0, sizeof(RtLevelReport) * RT_MAX_LEVELS);
		HIPCHK(hipHostGetDevicePointer((void**)&ln.reportDev, ln.report, 0));
		for (auto& e : ln.ev) HIPCHK(hipEventCreate(&e));
		if (ownStream) HIPCHK(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
		return RT_OK;
	}
	int inFlight() const { int n = 0; for (const Job& j : jobs) if (j.id && j.remaining > 0) n++; return n; }
	// Waits until the lanes are idle AND the GPU has finished what they queued (jobs may still be un-waited-for); call with
	// mtx held.  A worker is done with a shard once it has ENQUEUED it: the tail of the last level (shadow kernels, texel fix,
	// resolve) may still be running on the lane's non-blocking stream, which neither the NULL stream nor another lane's
	// stream is ordered behind -- so whoever goes on to use lane 0's queues or the shared accumulators waits for the
	// shards' end events as well.  (Events of retired jobs are complete: rt_render_wait synchronised them.)
	void drainLocked(std::unique_lock<std::mutex>& g) {
		cvDone.wait(g, [&] { return inFlight() == 0; });
		for (const Job& j : jobs)
			if (j.id)
				for (int k = 0; k < j.nShards; k++) (void)hipEventSynchronize(j.evDone[k]);
	}
	void drain() { std::unique_lock<std::mutex> g(mtx); drainLocked(g); }
	int ensureLaneAccum(int k, uint32_t npix) {
		RtAccum& a = laneAccum[k];
		if (a.npix >= npix && a.rgb) return RT_OK;
		if (a.rgb) { (void)hipFree(a.rgb); (void)hipFree(a.nonfinite); a.rgb = nullptr; a.nonfinite = nullptr; a.npix = 0; }
		HIPCHK(hipMalloc((void**)&a.rgb, (size_t)npix * 3 * sizeof(unsigned long long)));
		HIPCHK(hipMalloc((void**)&a.nonfinite, (size_t)npix));
		HIPCHK(hipMemset(a.rgb, 0, (size_t)npix * 3 * sizeof(unsigned long long)));
		HIPCHK(hipMemset(a.nonfinite, 0, (size_t)npix));
		a.npix = npix;
		return RT_OK;
	}
	int ensureAccum(uint32_t npix) {
		if (accum.npix >= npix && accum.rgb) return RT_OK;
		if (accum.rgb) { (void)hipFree(accum.rgb); (void)hipFree(accum.nonfinite); accum.rgb = nullptr; accum.nonfinite = nullptr; accum.npix = 0; }
		if (texHeads) { (void)hipFree(texHeads); texHeads = nullptr; }
		if (texelFix) {
			HIPCHK(hipMalloc((void**)&texHeads, (size_t)npix * sizeof(uint32_t)));
			HIPCHK(hipMemset(texHeads, 0xff, (size_t)npix * sizeof(uint32_t)));   // every list empty
		}
		HIPCHK(hipMalloc((void**)&accum.rgb, (size_t)npix * 3 * sizeof(unsigned long long)));
		HIPCHK(hipMalloc((void**)&accum.nonfinite, (size_t)npix));
		HIPCHK(hipMemset(accum.rgb, 0, (size_t)npix * 3 * sizeof(unsigned long long)));
		HIPCHK(hipMemset(accum.nonfinite, 0, (size_t)npix));
		accum.npix = npix;
		return RT_OK;
	}
};

namespace {

int treeDepth(const rt_flat_scene* s) {
	if (s->root_child < 0) return 0;
	int maxDepth = 0;
	std::vector<std::pair<int32_t, int>> st;
	st.push_back({ s->root_child, 1 });
	while (!st.empty()) {
		auto [n, d] = st.back();
		st.pop_back();
		maxDepth = std::max(maxDepth, d);
		for (int k = 0; k < 2; k++)
			if (s->bvh_nodes[n].child[k] >= 0) st.push_back({ s->bvh_nodes[n].child[k], d + 1 });
	}
	return maxDepth;
}

// A leaf whose primitives are consecutive triangles: (first index in tris[], count); count 0 otherwise.
struct TriRun { uint32_t first = 0, count = 0; };
TriRun leafRun(const rt_flat_scene* s, int32_t leafRef) {
	TriRun r;
	for (int64_t at = ~leafRef;; at++) {
		const uint32_t e = s->leaf_prims[at];
		const uint32_t ix = RT_PRIM_INDEX(e & ~RT_PRIM_LAST);
		if (RT_PRIM_TYPE(e) != RT_PRIM_TRI || (r.count > 0 && ix != r.first + r.count)) return TriRun{};
		if (r.count == 0) r.first = ix;
		r.count++;
		if (e & RT_PRIM_LAST) break;
	}
	return r;
}
// aux word of an external run (rt_kernels.h): short form for up to 63 triangles; long form (the count alone) when the
// run's triangles sit at the same indices in tris[] as their entries in leaf_prims[], as in a triangles-only scene
uint32_t externalAux(const TriRun& r, int32_t leafRef) {
	if (r.count == 0) return RT_TREELET_NO_RUN;
	if (r.count <= RT_TREELET_RUN_MAX && r.first < RT_TREELET_EMBED - RT_TREELET_RUN_MAX) return r.count << 26 | r.first;
	if (r.first == (uint32_t)~leafRef && r.count < RT_TREELET_EMBED) return r.count;
	return RT_TREELET_NO_RUN;
}

// The tree the packet kernels walk: the reference's nodes, plus -- below every leaf of more than RT_TREELET_EMBED_MAX
// consecutive triangles -- a small tree over that leaf's triangles.  The reference's build stops at depth 16
// (bvh.h:6), which leaves ~76 triangles per leaf in a 5 M-triangle mesh, and it tests all of them for every ray that
// enters the leaf's box.  What it returns from a leaf is a pure function of the triangles' own hit tests -- the
// nearest t above 1e-5, first in leaf order among equals; for shadows the product over all triangles hit before the
// light -- so any way of skipping triangles that certainly cannot be hit returns the same.  The sub-tree's boxes are
// therefore CONSERVATIVE (slightly inflated, and a test too close to call counts as a hit, RT_TREELET_CONS), unlike
// the reference's own boxes above them, whose exact floating-point verdicts are part of its result.
struct AugTree {
	struct Node { float box[2][6]; int32_t child[2]; bool cons; };   // child >= 0: node; < 0: ~index into leaves
	struct Leaf { int32_t ref; bool sub; uint32_t first, count; };   // !sub: the reference leaf `ref` (run first/count, count 0 = not a run); sub: subTris[first .. first+count)
	std::vector<Node> nodes;
	std::vector<Leaf> leaves;
	std::vector<uint32_t> subTris;   // triangle indices
	std::vector<uint32_t> subKeys;   // ... and their tie-break keys (position in leaf_prims + 1)
	int32_t root = -1;
	int depth = 0;
};

void triBounds(const rt_tri& t, float lo[3], float hi[3]) {
	for (int a = 0; a < 3; a++) { lo[a] = std::min({ t.p0[a], t.p1[a], t.p2[a] }); hi[a] = std::max({ t.p0[a], t.p1[a], t.p2[a] }); }
}
// bounds of subTris[first, first+count), inflated so that no hit the triangle test accepts can fall outside
void subBox(const rt_flat_scene* s, const AugTree& A, uint32_t first, uint32_t count, float box[6]) {
	float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
	for (uint32_t i = 0; i < count; i++) {
		float l[3], h[3];
		triBounds(s->tris[A.subTris[first + i]], l, h);
		for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
	}
	for (int a = 0; a < 3; a++) {
		const float pad = 1e-5f * std::max({ fabsf(lo[a]), fabsf(hi[a]), hi[a] - lo[a] }) + 1e-6f;
		box[2 * a] = lo[a] - pad; box[2 * a + 1] = hi[a] + pad;
	}
}
// The sub-tree over subTris[first, first+count) of one fat leaf, built into vectors of its own (nodes and leaves numbered
// from 0) so that the fat leaves of a scene -- 65 536 of them under C5's depth-capped tree -- can be built on all host cores;
// each touches only its own range of subTris / subKeys.  Median split on the longest axis of the centroid bounds.
struct SubTree { std::vector<AugTree::Node> nodes; std::vector<AugTree::Leaf> leaves; int maxDepth = 0; int32_t root = 0; };
int32_t buildSub(const rt_flat_scene* s, AugTree& A, SubTree& T, uint32_t first, uint32_t count, int depth) {
	T.maxDepth = std::max(T.maxDepth, depth);
	if (count <= RT_TREELET_SUBLEAF_MAX) {
		T.leaves.push_back({ 0, true, first, count });
		return ~(int32_t)(T.leaves.size() - 1);
	}
	float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
	auto centroid = [&](uint32_t tri, int a) { const rt_tri& t = s->tris[tri]; return (t.p0[a] + t.p1[a] + t.p2[a]) * (1.0f / 3.0f); };
	for (uint32_t i = 0; i < count; i++)
		for (int a = 0; a < 3; a++) { const float c = centroid(A.subTris[first + i], a); lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c); }
	int axis = 0;
	if (hi[1] - lo[1] > hi[axis] - lo[axis]) axis = 1;
	if (hi[2] - lo[2] > hi[axis] - lo[axis]) axis = 2;
	// order triangle indices and keys together
	std::vector<std::pair<uint32_t, uint32_t>> tmp(count);
	for (uint32_t i = 0; i < count; i++) tmp[i] = { A.subTris[first + i], A.subKeys[first + i] };
	const uint32_t half = count / 2;
	std::nth_element(tmp.begin(), tmp.begin() + half, tmp.end(), [&](const auto& x, const auto& y) {
		const float cx = centroid(x.first, axis), cy = centroid(y.first, axis);
		return cx < cy || (cx == cy && x.second < y.second);
	});
	for (uint32_t i = 0; i < count; i++) { A.subTris[first + i] = tmp[i].first; A.subKeys[first + i] = tmp[i].second; }
	const int32_t me = (int32_t)T.nodes.size();
	T.nodes.push_back({});
	AugTree::Node n{};
	n.cons = true;
	subBox(s, A, first, half, n.box[0]);
	subBox(s, A, first + half, count - half, n.box[1]);
	n.child[0] = buildSub(s, A, T, first, half, depth + 1);
	n.child[1] = buildSub(s, A, T, first + half, count - half, depth + 1);
	T.nodes[(size_t)me] = n;
	return me;
}

void buildAugTree(const rt_flat_scene* s, bool split, AugTree& A) {
	A = AugTree{};
	if (s->root_child < 0) return;
	A.root = s->root_child;
	A.nodes.resize((size_t)s->n_bvh_nodes);
	// depths of the reference nodes (children follow their parents in the pre-order array)
	std::vector<int> depthOf((size_t)s->n_bvh_nodes, 0);
	depthOf[(size_t)s->root_child] = 1;
	// pass 1: the fat leaves and their triangles
	struct Job { uint32_t first, count; int depth; SubTree tree; };
	std::vector<Job> jobs;
	for (int32_t i = 0; i < s->n_bvh_nodes; i++) {
		const rt_bvh_node& b = s->bvh_nodes[i];
		for (int k = 0; k < 2; k++) {
			const int32_t c = b.child[k];
			if (c >= 0) { depthOf[(size_t)c] = depthOf[(size_t)i] + 1; continue; }
			const TriRun r = leafRun(s, c);
			if (split && r.count > RT_TREELET_EMBED_MAX) {
				jobs.push_back({ (uint32_t)A.subTris.size(), r.count, depthOf[(size_t)i] + 1, {} });
				for (uint32_t j = 0; j < r.count; j++) { A.subTris.push_back(r.first + j); A.subKeys.push_back((uint32_t)~c + j + 1u); }
			}
		}
	}
	// pass 2: their sub-trees, independent of each other
	{
		const unsigned nThreads = jobs.size() < 64 ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
		std::vector<std::thread> pool;
		for (unsigned t = 0; t < nThreads; t++)
			pool.emplace_back([&, t] {
				for (size_t j = t; j < jobs.size(); j += nThreads) jobs[j].tree.root = buildSub(s, A, jobs[j].tree, jobs[j].first, jobs[j].count, jobs[j].depth);
			});
		for (auto& th : pool) th.join();
	}
	// pass 3: the reference nodes, with the sub-trees spliced in where their leaves were (same numbering as a sequential build)
	size_t next = 0;
	for (int32_t i = 0; i < s->n_bvh_nodes; i++) {
		const rt_bvh_node& b = s->bvh_nodes[i];
		AugTree::Node n{};
		memcpy(n.box, b.box, sizeof n.box);
		n.cons = false;
		for (int k = 0; k < 2; k++) {
			const int32_t c = b.child[k];
			if (c >= 0) { n.child[k] = c; continue; }
			const TriRun r = leafRun(s, c);
			if (split && r.count > RT_TREELET_EMBED_MAX) {
				const SubTree& T = jobs[next++].tree;
				const int32_t nodeBase = (int32_t)A.nodes.size(), leafBase = (int32_t)A.leaves.size();
				auto global = [&](int32_t ref) { return ref >= 0 ? nodeBase + ref : ~(leafBase + ~ref); };
				for (AugTree::Node m : T.nodes) { m.child[0] = global(m.child[0]); m.child[1] = global(m.child[1]); A.nodes.push_back(m); }
				A.leaves.insert(A.leaves.end(), T.leaves.begin(), T.leaves.end());
				n.child[k] = global(T.root);
				A.depth = std::max(A.depth, T.maxDepth);
			} else {
				A.leaves.push_back({ c, false, r.first, r.count });
				n.child[k] = ~(int32_t)(A.leaves.size() - 1);
				A.depth = std::max(A.depth, depthOf[(size_t)i] + 1);
			}
		}
		A.nodes[(size_t)i] = n;
	}
}

// Trees for Mesh::rayhit (meshHit, rt_device.h): for every mesh that can be asked for a hit on its own -- a CSG leaf, or
// a mesh whose material transmits (Object::transmitRay looks for the exit on the mesh) -- and has enough triangles for
// a tree to pay.  Median split over triangle centroids, conservative (inflated) boxes, leaves of up to 4 triangles
// listed as offsets into the mesh's triangles.
struct MeshTrees {
	std::vector<rt_bvh_node> nodes;
	std::vector<int32_t> order, roots;
	int depth = 0;
};
void meshBox(const rt_tri* tris, const int32_t* idx, int count, float box[6]) {
	float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
	for (int i = 0; i < count; i++) {
		float l[3], h[3];
		triBounds(tris[idx[i]], l, h);
		for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
	}
	for (int a = 0; a < 3; a++) {
		const float pad = 1e-5f * std::max({ fabsf(lo[a]), fabsf(hi[a]), hi[a] - lo[a] }) + 1e-6f;
		box[2 * a] = lo[a] - pad; box[2 * a + 1] = hi[a] + pad;
	}
}
// builds the subtree over order[first, first+count) (count > 4); returns its node index
int32_t buildMeshNode(const rt_tri* tris, MeshTrees& M, int first, int count, int depth) {
	M.depth = std::max(M.depth, depth);
	int32_t* idx = M.order.data() + first;
	float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
	auto centroid = [&](int32_t i, int a) { const rt_tri& t = tris[i]; return (t.p0[a] + t.p1[a] + t.p2[a]) * (1.0f / 3.0f); };
	for (int i = 0; i < count; i++)
		for (int a = 0; a < 3; a++) { const float c = centroid(idx[i], a); lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c); }
	int axis = 0;
	if (hi[1] - lo[1] > hi[axis] - lo[axis]) axis = 1;
	if (hi[2] - lo[2] > hi[axis] - lo[axis]) axis = 2;
	const int half = count / 2;
	std::nth_element(idx, idx + half, idx + count, [&](int32_t x, int32_t y) {
		const float cx = centroid(x, axis), cy = centroid(y, axis);
		return cx < cy || (cx == cy && x < y);
	});
	const int32_t me = (int32_t)M.nodes.size();
	M.nodes.push_back(rt_bvh_node{});
	rt_bvh_node n{};
	const int part[2][2] = { { first, half }, { first + half, count - half } };
	for (int k = 0; k < 2; k++) {
		meshBox(tris, M.order.data() + part[k][0], part[k][1], n.box[k]);
		if (part[k][1] <= 4) { n.child[k] = ~part[k][0]; n.count[k] = part[k][1]; }
		else { n.child[k] = buildMeshNode(tris, M, part[k][0], part[k][1], depth + 1); n.count[k] = 0; }
	}
	M.nodes[(size_t)me] = n;
	return me;
}
void buildMeshTrees(const rt_flat_scene* s, MeshTrees& M) {
	M = MeshTrees{};
	M.roots.assign((size_t)std::max(s->n_meshes, 1), -1);
	if (getenv("RT_NO_MESH_TREES")) return;
	std::vector<char> wanted((size_t)s->n_meshes, 0);
	for (int32_t i = 0; i < s->n_csg_nodes; i++)
		if (s->csg_nodes[i].kind == RT_CSG_MESH && s->csg_nodes[i].mesh >= 0 && s->csg_nodes[i].mesh < s->n_meshes) wanted[(size_t)s->csg_nodes[i].mesh] = 1;
	for (int32_t i = 0; i < s->n_meshes; i++) {
		const float* kt = s->materials[s->meshes[i].mat].transmissive;
		if (kt[0] != 0 || kt[1] != 0 || kt[2] != 0) wanted[(size_t)i] = 1;
	}
	for (int32_t i = 0; i < s->n_meshes; i++) {
		const rt_mesh& m = s->meshes[i];
		if (!wanted[(size_t)i] || m.tri_count < 32) continue;
		const int first = (int)M.order.size();
		for (int j = 0; j < m.tri_count; j++) M.order.push_back(j);
		M.roots[(size_t)i] = buildMeshNode(s->mesh_tris + m.tri_first, M, first, m.tri_count, 1);
	}
	if (M.depth > RT_MESH_STACK) { M.nodes.clear(); M.order.clear(); M.roots.assign(M.roots.size(), -1); }   // (a degenerate mesh: keep the scan)
}

// ---------------------------------------------------------------------------------------------------------
// Cutting a tree into treelets, on several threads.  A treelet is filled breadth first from its root until its slots are used up; the
// nodes that no longer fit root treelets of their own, one level further down.  What a treelet holds depends on its root alone, so once
// the top of the tree is cut and enough roots are waiting, the subtrees below them are cut independently -- each into treelets numbered
// from 0, with the references between them local -- and then copied behind the top, their references moved by where they landed.  The
// treelets come out grouped by subtree instead of level by level; every treelet's contents are what the sequential cut gives it.
// ---------------------------------------------------------------------------------------------------------
int g_cutThreads = 0;   // rt_debug_treelet_digest: the number of threads to cut with (0: by the size of the tree)
struct CutRoot { int32_t node; int level; int depth; };
struct Cut {
	std::vector<rt_bvh_node> slots;   // treelets 0 .. roots.size() - 1 as far as they have been cut
	std::vector<uint8_t> records;     // per cut treelet: the node records at its front (the rest of its slots are triangles)
	std::vector<CutRoot> roots;       // roots[t]: what treelet t starts from; a reference to treelet t is t * slotsPerTreelet (| flags)
	int levels = 0, depth = 0;
	bool externalRuns = false;
	std::string error;
};
// one(cut, t, scratch): cuts treelet t of `cut` (appends the roots it leaves, references them by their index in cut.roots); false = failed
// reloc(record, f): replaces the index part i of every reference to a node record in `record` by f(i)
template <class Scratch, class One, class Reloc>
int cutTree(int32_t root, size_t slotsPerTreelet, size_t slotsPerRecord, size_t treeNodes, One one, Reloc reloc, rt::PodVec<rt_bvh_node>& out, bool& externalRuns, int& depth) {
	out.clear();
	externalRuns = false; depth = 0;
	if (root < 0) return 0;
	static const unsigned maxThreads = std::max(1u, std::min(envU32("RT_UPLOAD_THREADS", 16), std::max(1u, std::thread::hardware_concurrency())));
	const unsigned threads = g_cutThreads > 0 ? (unsigned)g_cutThreads : treeNodes >= 20000 ? maxThreads : 1u;
	Cut top;
	top.roots.push_back({ root, 0, 1 });
	top.slots.reserve(threads > 1 ? 4096 * slotsPerTreelet : (treeNodes / 2 + 64) * slotsPerTreelet);
	Scratch scratch;
	size_t t = 0;
	for (; t < top.roots.size(); t++) {
		if (threads > 1 && t >= 64 && top.roots.size() - t >= 16 * (size_t)threads) break;   // enough subtrees waiting: the rest in parallel
		if (!one(top, t, scratch)) { rt::set_error(top.error); return -1; }
	}
	const size_t T0 = t, nJobs = top.roots.size() - T0;
	std::vector<Cut> jobs(nJobs);
	if (nJobs > 0) {
		std::atomic<size_t> next{ 0 };
		std::vector<std::thread> pool;
		for (unsigned w = 0; w < threads; w++)
			pool.emplace_back([&] {
				Scratch mine;
				for (size_t j; (j = next.fetch_add(1)) < nJobs;) {
					Cut& C = jobs[j];
					C.roots.push_back(top.roots[T0 + j]);
					for (size_t lt = 0; lt < C.roots.size(); lt++)
						if (!one(C, lt, mine)) { C.roots.clear(); break; }   // (error kept in C.error)
				}
			});
		for (auto& th : pool) th.join();
	}
	std::vector<size_t> base(nJobs + 1, T0);
	for (size_t j = 0; j < nJobs; j++) {
		if (jobs[j].roots.empty()) { rt::set_error(jobs[j].error); return -1; }
		base[j + 1] = base[j] + jobs[j].roots.size();
	}
	const size_t total = base[nJobs];
	out.resize(total * slotsPerTreelet);
	memcpy(out.data(), top.slots.data(), T0 * slotsPerTreelet * sizeof(rt_bvh_node));
	// the top's references to the waiting roots: root T0 + j became treelet base[j]
	for (size_t tt = 0; tt < T0; tt++)
		for (size_t r = 0; r < top.records[tt]; r++)
			reloc(&out[tt * slotsPerTreelet + r * slotsPerRecord], [&](size_t i) {
				const size_t ti = i / slotsPerTreelet;
				return ti >= T0 ? base[ti - T0] * slotsPerTreelet + i % slotsPerTreelet : i;
			});
	int levels = top.levels;
	depth = top.depth; externalRuns = top.externalRuns;
	if (nJobs > 0) {
		std::atomic<size_t> next{ 0 };
		std::vector<std::thread> pool;
		for (unsigned w = 0; w < threads; w++)
			pool.emplace_back([&] {
				for (size_t j; (j = next.fetch_add(1)) < nJobs;) {
					const Cut& C = jobs[j];
					rt_bvh_node* dst = out.data() + base[j] * slotsPerTreelet;
					memcpy(dst, C.slots.data(), C.roots.size() * slotsPerTreelet * sizeof(rt_bvh_node));
					const size_t shift = base[j] * slotsPerTreelet;
					for (size_t lt = 0; lt < C.roots.size(); lt++)
						for (size_t r = 0; r < C.records[lt]; r++) reloc(dst + lt * slotsPerTreelet + r * slotsPerRecord, [&](size_t i) { return i + shift; });
				}
			});
		for (auto& th : pool) th.join();
		for (const Cut& C : jobs) { levels = std::max(levels, C.levels); depth = std::max(depth, C.depth); externalRuns |= C.externalRuns; }
	}
	return levels;
}

// Cuts the augmented tree into treelets (layout in rt_kernels.h).  Returns the number of treelet levels.
int buildTreelets(const rt_flat_scene* s, const AugTree& A, bool embed, rt::PodVec<rt_bvh_node>& out, bool& externalRuns) {
	struct Member { int32_t node; bool embedLeaf[2]; };
	struct Scratch { std::vector<Member> member; std::vector<int32_t> childSlot; };   // childSlot: per member, per child: slot of an in-treelet inner child, -1 otherwise
	// triangles a leaf brings into the treelet if it is embedded: sub-leaves always, reference runs up to RT_TREELET_EMBED_MAX
	auto embeddable = [&](int32_t c) -> uint32_t {
		const AugTree::Leaf& L = A.leaves[(size_t)~c];
		if (L.sub) return L.count;
		return (embed && L.count > 0 && L.count <= RT_TREELET_EMBED_MAX && L.first < RT_TREELET_EMBED) ? L.count : 0u;
	};
	auto one = [&](Cut& C, size_t t, Scratch& X) -> bool {
		std::vector<Member>& member = X.member;
		std::vector<int32_t>& childSlot = X.childSlot;
		C.slots.resize((t + 1) * RT_TREELET_SLOTS, rt_bvh_node{});
		const int level = C.roots[t].level;
		C.levels = std::max(C.levels, level + 1);
		member.clear();
		uint32_t used = 0;
		auto join = [&](int32_t node, bool must) {
			const AugTree::Node& b = A.nodes[(size_t)node];
			uint32_t tris[2] = { 0, 0 };
			for (int k = 0; k < 2; k++) if (b.child[k] < 0) tris[k] = embeddable(b.child[k]);
			if (!must && used + 1 + tris[0] + tris[1] > RT_TREELET_SLOTS) return false;
			Member m{ node, { false, false } };
			used += 1;
			for (int k = 0; k < 2; k++)   // the root joins whatever it brings; its runs are embedded while they fit (sub-leaves always do)
				if (tris[k] && used + tris[k] <= RT_TREELET_SLOTS) { m.embedLeaf[k] = true; used += tris[k]; }
			member.push_back(m);
			return true;
		};
		join(C.roots[t].node, true);
		childSlot.clear();
		for (size_t i = 0; i < member.size(); i++) {
			const AugTree::Node& b = A.nodes[(size_t)member[i].node];
			for (int k = 0; k < 2; k++) {
				int32_t slot = -1;
				if (b.child[k] >= 0 && join(b.child[k], false)) slot = (int32_t)member.size() - 1;
				childSlot.push_back(slot);
			}
		}
		C.records.push_back((uint8_t)member.size());
		uint32_t triSlot = (uint32_t)member.size();
		for (size_t i = 0; i < member.size(); i++) {
			const AugTree::Node& b = A.nodes[(size_t)member[i].node];
			rt_bvh_node& o = C.slots[t * RT_TREELET_SLOTS + i];
			memcpy(o.box, b.box, sizeof o.box);
			for (int k = 0; k < 2; k++) {
				const int32_t c = b.child[k];
				if (c < 0) {
					const AugTree::Leaf& L = A.leaves[(size_t)~c];
					o.child[k] = L.sub ? -1 : L.ref;
					if (member[i].embedLeaf[k]) {
						o.count[k] = (int32_t)(L.count << 26 | RT_TREELET_EMBED | (uint32_t)level << 4 | triSlot);
						for (uint32_t j = 0; j < L.count; j++, triSlot++) {
							const uint32_t index = L.sub ? A.subTris[L.first + j] : L.first + j;
							rt_tri tri = s->tris[index];
							tri.mesh = (int32_t)index;                                                             // -> T.best
							tri.uv = (int32_t)(L.sub ? A.subKeys[L.first + j] : (uint32_t)~L.ref + j + 1u);      // -> tie-break key
							memcpy(&C.slots[t * RT_TREELET_SLOTS + triSlot], &tri, sizeof tri);
						}
					} else {
						if (L.sub) { C.error = "internal: a sub-leaf did not fit its treelet"; return false; }
						o.count[k] = (int32_t)externalAux(TriRun{ L.first, L.count }, L.ref);
						C.externalRuns |= (uint32_t)o.count[k] != RT_TREELET_NO_RUN;
					}
				} else {
					const uint32_t flag = A.nodes[(size_t)c].cons ? RT_TREELET_CONS : 0u;
					if (childSlot[2 * i + k] >= 0) o.child[k] = (int32_t)((uint32_t)(t * RT_TREELET_SLOTS + childSlot[2 * i + k]) | flag);
					else {
						o.child[k] = (int32_t)((uint32_t)(C.roots.size() * RT_TREELET_SLOTS) | flag);
						C.roots.push_back({ c, level + 1, 0 });
					}
					o.count[k] = 0;
				}
			}
		}
		return true;
	};
	auto reloc = [](rt_bvh_node* rec, auto&& f) {
		for (int k = 0; k < 2; k++) {
			const int32_t c = rec->child[k];
			if (c >= 0) rec->child[k] = (int32_t)((uint32_t)f((size_t)((uint32_t)c & (RT_TREELET_CONS - 1u))) | ((uint32_t)c & RT_TREELET_CONS));
		}
	};
	int depth = 0;
	return cutTree<Scratch>(A.root, RT_TREELET_SLOTS, 1, A.nodes.size(), one, reloc, out, externalRuns, depth);
}

// The same tree as quad nodes for the packet walks (layout in rt_kernels.h, reasons in rt_kernels.hip "Quad nodes"): the node that
// stands for binary node n lists, for each child of n, that child's two children with their boxes -- or the child itself with its own
// box where it is a leaf, or where it roots a sub-tree of the other regime (n a reference node, the child the conservative tree under
// a fat leaf: the reference's exact verdict on the leaf's box must be kept).  A quad takes two 64 B slots of a treelet; the treelets
// are cut as above (breadth first, a quad joins only if the triangles of its embeddable leaves fit as well).  Returns the number of
// treelet levels, `quadDepth` = quads on the longest root-to-leaf path.
int buildQuadTreelets(const rt_flat_scene* s, const AugTree& A, bool embed, rt::PodVec<rt_bvh_node>& out, bool& externalRuns, int& quadDepth) {
	struct Entry { float box[6]; int32_t ref; };   // ref >= 0: binary node whose quad this entry leads to; < 0: ~index into A.leaves
	static const uint32_t collapse = envU32("RT_QUAD_COLLAPSE", 3);   // diagnostic: bit 0 reference nodes, bit 1 sub-tree nodes
	auto entriesOf = [&](int32_t n, Entry* e, bool skipLevel) -> int {
		const AugTree::Node& b = A.nodes[(size_t)n];
		int count = 0;
		for (int k = 0; k < 2; k++) {
			const int32_t c = b.child[k];
			if (skipLevel && c >= 0 && A.nodes[(size_t)c].cons == b.cons && (collapse & (b.cons ? 2u : 1u))) {   // same regime: the child's own box is skipped
				const AugTree::Node& m = A.nodes[(size_t)c];
				for (int j = 0; j < 2; j++) { memcpy(e[count].box, m.box[j], sizeof e[count].box); e[count].ref = m.child[j]; count++; }
			} else { memcpy(e[count].box, b.box[k], sizeof e[count].box); e[count].ref = c; count++; }
		}
		return count;
	};
	struct Member { int32_t node; int depth; Entry e[4]; int n; bool embedLeaf[4]; int32_t childSlot[4]; };
	struct Scratch { std::vector<Member> member; };
	auto embeddable = [&](int32_t c) -> uint32_t {
		const AugTree::Leaf& L = A.leaves[(size_t)~c];
		if (L.sub) return L.count;
		return (embed && L.count > 0 && L.count <= RT_TREELET_EMBED_MAX && L.first < RT_TREELET_EMBED) ? L.count : 0u;
	};
	auto one = [&](Cut& C, size_t t, Scratch& X) -> bool {
		std::vector<Member>& member = X.member;
		C.slots.resize((t + 1) * RT_QUAD_SLOTS, rt_bvh_node{});
		const int level = C.roots[t].level;
		C.levels = std::max(C.levels, level + 1);
		member.clear();
		uint32_t used = 0;
		auto join = [&](int32_t node, int depth, bool must) {
			Member m{};
			m.node = node; m.depth = depth;
			uint32_t tris[4] = { 0, 0, 0, 0 }, all = 0, mustEmbed = 0;
			for (int pass = 0; pass < 2; pass++) {
				m.n = entriesOf(node, m.e, pass == 0);
				all = 0; mustEmbed = 0;
				for (int i = 0; i < 4; i++) tris[i] = 0;
				for (int i = 0; i < m.n; i++) if (m.e[i].ref < 0) { tris[i] = embeddable(m.e[i].ref); all += tris[i]; if (A.leaves[(size_t)~m.e[i].ref].sub) mustEmbed += tris[i]; }
				// sub-leaves have to sit in their quad's treelet: four of them with four triangles each do not fit beside the quad, so such a
				// node keeps its two children as entries (each then a quad of two sub-leaves)
				if (2 + mustEmbed <= RT_QUAD_SLOTS) break;
			}
			if (!must && used + 2 + all > RT_QUAD_SLOTS) return false;
			used += 2;
			for (int i = 0; i < m.n; i++) {
				m.childSlot[i] = -1;
				m.embedLeaf[i] = tris[i] && used + tris[i] <= RT_QUAD_SLOTS;   // (the root joins whatever it brings; its runs are embedded while they fit)
				if (m.embedLeaf[i]) used += tris[i];
			}
			C.depth = std::max(C.depth, depth);
			member.push_back(m);
			return true;
		};
		join(C.roots[t].node, C.roots[t].depth, true);
		for (size_t i = 0; i < member.size(); i++)
			for (int k = 0; k < member[i].n; k++) {
				const int32_t ref = member[i].e[k].ref;
				const int depth = member[i].depth + 1;
				if (ref >= 0 && join(ref, depth, false)) member[i].childSlot[k] = 2 * ((int32_t)member.size() - 1);
			}
		C.records.push_back((uint8_t)member.size());
		uint32_t triSlot = 2u * (uint32_t)member.size();
		for (size_t i = 0; i < member.size(); i++) {
			const Member& m = member[i];
			float* rec = reinterpret_cast<float*>(&C.slots[t * RT_QUAD_SLOTS + 2 * i]);   // 32 floats: 4 boxes, 4 children, 4 aux words
			int32_t* child = reinterpret_cast<int32_t*>(rec + 24);
			uint32_t* aux = reinterpret_cast<uint32_t*>(rec + 28);
			for (int k = 0; k < 4; k++) {
				if (k >= m.n) {   // no such entry: a box nothing passes
					for (int a = 0; a < 3; a++) { rec[6 * k + 2 * a] = 3.0e38f; rec[6 * k + 2 * a + 1] = -3.0e38f; }
					child[k] = RT_QUAD_EMPTY; aux[k] = 0;
					continue;
				}
				memcpy(rec + 6 * k, m.e[k].box, 6 * sizeof(float));
				const int32_t c = m.e[k].ref;
				if (c < 0) {
					const AugTree::Leaf& L = A.leaves[(size_t)~c];
					child[k] = L.sub ? -1 : L.ref;
					if (m.embedLeaf[k]) {
						aux[k] = L.count << 26 | RT_TREELET_EMBED | (uint32_t)level << RT_QUAD_SLOT_BITS | triSlot;
						for (uint32_t j = 0; j < L.count; j++, triSlot++) {
							const uint32_t index = L.sub ? A.subTris[L.first + j] : L.first + j;
							rt_tri tri = s->tris[index];
							tri.mesh = (int32_t)index;                                                             // -> T.best
							tri.uv = (int32_t)(L.sub ? A.subKeys[L.first + j] : (uint32_t)~L.ref + j + 1u);      // -> tie-break key
							memcpy(&C.slots[t * RT_QUAD_SLOTS + triSlot], &tri, sizeof tri);
						}
					} else {
						if (L.sub) { C.error = "internal: a sub-leaf did not fit its treelet"; return false; }
						aux[k] = externalAux(TriRun{ L.first, L.count }, L.ref);
						C.externalRuns |= aux[k] != RT_TREELET_NO_RUN;
					}
				} else {
					const uint32_t flag = A.nodes[(size_t)c].cons ? RT_TREELET_CONS : 0u;
					// (aux of an inner entry = the treelet level of the quad it leads to: which of the wave's LDS buffers holds, or will hold, that treelet)
					if (m.childSlot[k] >= 0) { child[k] = (int32_t)((uint32_t)(t * RT_QUAD_SLOTS + (size_t)m.childSlot[k]) | flag); aux[k] = (uint32_t)level; }
					else {   // roots another treelet, one level further down
						child[k] = (int32_t)((uint32_t)(C.roots.size() * RT_QUAD_SLOTS) | flag);
						C.roots.push_back({ c, level + 1, m.depth + 1 });
						aux[k] = (uint32_t)level + 1u;
					}
				}
			}
		}
		return true;
	};
	auto reloc = [](rt_bvh_node* rec, auto&& f) {
		int32_t* child = reinterpret_cast<int32_t*>(reinterpret_cast<float*>(rec) + 24);
		for (int k = 0; k < 4; k++) {
			const int32_t c = child[k];
			if (c >= 0 && c != RT_QUAD_EMPTY) child[k] = (int32_t)((uint32_t)f((size_t)((uint32_t)c & (RT_TREELET_CONS - 1u))) | ((uint32_t)c & RT_TREELET_CONS));
		}
	};
	return cutTree<Scratch>(A.root, RT_QUAD_SLOTS, 2, A.nodes.size(), one, reloc, out, externalRuns, quadDepth);
}

// Where a call's rays come from and where first-level results go.
struct Source {
	bool camera = false;
	rt_camera cam{};
	int W = 0, H = 0, aaX = 1, aaY = 1, tileRank = 0, tileRanks = 1;
	const float* a = nullptr; const float* b = nullptr;   // device arrays for ray batches (origin, target)
	uint8_t* outHit = nullptr; int32_t* outId = nullptr; float* outT = nullptr;
};

struct Run {
	rt_scene* sc; rt_scene::Lane* ln; hipStream_t st; Source src; int maxDepth; uint32_t seed; uint32_t flags; uint32_t aaSq;
	rt_stats stats{};
	RtPick pick{ nullptr, nullptr, 0 };   // rt_trace_pick: where the narration's events go
	uint32_t texRecords = 0;               // RtTexFix records the current batch has left so far
	uint32_t stateCases = 0;               // failed exit searches since begin() (the lane's counter is cumulative over the call's batches)
	RtAccum acc{};                         // the accumulators this run adds into: the scene's shared ones, or the lane's own (RT_FLAG_WHOLE_FRAME); set by use()
	Run& use(const RtAccum& a) { acc = a; return *this; }

	// Which traversal form a level uses.  Level 0 of a frame (and its shadow rays) is coherent by construction: 64
	// consecutive queue slots are an 8x8 pixel block.  Deeper levels and caller-supplied rays are walked per lane.
	// RT_PACKET_LEVELS > 1 sends deeper levels through the packet kernels as well: 64 consecutive children in the order the shading
	// kernel compacted them, i.e. still neighbours in the image.  Measured (profiles/r02_deeper_levels.txt) and not the default: the
	// bundled scenes move by -8 .. +5 % (i, a mesh under two spheres: 0.36 -> 0.34 ms), the CSG lattice C3 loses (2.35 -> 2.57 ms with
	// level 1 in packets, 4.2 ms with all levels: a CSG leaf is evaluated by the lanes that reach it while the rest of the packet
	// waits), and a reflective 100 352-triangle height field at 1080p, whose reflected rays leave neighbouring triangles in different
	// directions, takes 4.3 ms per lane walk against 7.9 ms in packets.
	int packetLevels = (int)envU32("RT_PACKET_LEVELS", 1);
	// (RT_PACKET_BATCHES: caller-supplied rays go through the packet kernels too, 64 consecutive rays per packet -- what the tests use to
	// compare the two walks on the same rays; read per call)
	bool coherent(int level) const { return (src.camera || getenv("RT_PACKET_BATCHES") != nullptr) && level < packetLevels; }

	int syncCounters() {
		if (int rcL = launchesOk()) return rcL;
		HIPCHK(hipMemcpyAsync(ln->ctrHost, ln->ctr, sizeof(RtCounters), hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		return RT_OK;
	}

	// One batch of first-level rays [r0, r0+cnt) through all levels.  Returns 1 if a queue overflowed.
	int levels(uint32_t r0, uint32_t cnt) {
		const bool count = (flags & RT_FLAG_COUNTERS) != 0, timeK = (flags & RT_FLAG_TIME_KERNELS) != 0;
		int cur = 0;
		if (src.camera) rtk_generate(st, src.cam, src.W, src.H, src.aaX, src.aaY, src.tileRank, src.tileRanks, r0, cnt, ln->ray[0]);
		else rtk_load_rays(st, src.a + 3 * (size_t)r0, src.b + 3 * (size_t)r0, cnt, 1, ln->ray[0], r0);
		uint32_t n = cnt;
		bool shadowPending = false;
		uint64_t nearRays = 0, shadowRays = 0, nLevels = 0;
		texRecords = 0;
		for (int level = 0; level <= maxDepth && n > 0; level++) {
			HIPCHK(hipMemsetAsync(ln->ctr, 0, RT_COUNTER_RESET_BYTES, st));   // queue counts and work cursors
			if (timeK) HIPCHK(hipEventRecord(ln->ev[2], st));
			rtk_intersect(st, sc->dev, sc->hasCsg, count, coherent(level), coherent(level) ? sc->maxBlocksPacketNear : sc->maxBlocks, ln->ray[cur], n, ln->hitq, ln->ctr);
			if (timeK) HIPCHK(hipEventRecord(ln->ev[3], st));
			rtk_shade(st, sc->dev, sc->hasCsg, ln->ray[cur], ln->hitq, n, ln->ray[cur ^ 1], ln->surfq, ln->transq, ln->ctr, acc, aaSq, maxDepth,
			          sc->shadowRaysPerSurface > 0, level == 0 ? src.outHit : nullptr, src.outId, src.outT, pick, sc->texFix(*ln));
			if (sc->hasTransmissive) rtk_transmit(st, sc->dev, sc->hasCsg, ln->transq, n, ln->ray[cur ^ 1], ln->ctr, &ln->ctr->state_cases, acc, aaSq, maxDepth, pick);
			// Level 0 of a frame lit by point-type lights only: the shadow packets are queued before the host has seen the
			// counts (the kernels read them on the device), and the host waits for the count read-back while they run --
			// the lane's stream never idles through a host round trip.
			const bool early = coherent(level) && !timeK && !count && sc->areaLights.empty() && sc->nPointLights > 0 && sc->shadowRaysPerSurface > 0;
			if (early) {
				if (int rcL = launchesOk()) return rcL;
				HIPCHK(hipMemcpyAsync(ln->ctrHost, ln->ctr, sizeof(RtCounters), hipMemcpyDeviceToHost, st));
				HIPCHK(hipEventRecord(ln->ev[4], st));
				rtk_shadow_point(st, sc->dev, sc->hasCsg, false, true, sc->maxBlocksPacketShadow, ln->surfq, n, sc->pointLights, sc->nPointLights, acc, aaSq, ln->ctr, true, sc->maxBlocksPacketShadow2);
				HIPCHK(hipEventSynchronize(ln->ev[4]));
			} else
			if (int rc = syncCounters()) return rc;
			if (timeK) {
				float ms = 0;
				HIPCHK(hipEventElapsedTime(&ms, ln->ev[2], ln->ev[3])); stats.nearest_ms += ms;
				if (shadowPending) { HIPCHK(hipEventElapsedTime(&ms, ln->ev[4], ln->ev[5])); stats.shadow_ms += ms; shadowPending = false; }
			}
			texRecords = ln->ctrHost->texrec_count;
			stateCases = ln->ctrHost->state_cases;   // (cumulative since begin())
			if (ln->ctrHost->overflow) return 1;
			nLevels++;
			// inactive slots of a partial edge tile are not rays
			nearRays += n;
			const uint32_t nsurf = ln->ctrHost->surf_count;
			if (early) shadowRays += (uint64_t)nsurf * sc->shadowRaysPerSurface;
			else if (nsurf > 0 && sc->shadowRaysPerSurface > 0) {
				if (timeK) HIPCHK(hipEventRecord(ln->ev[4], st));
				if (sc->nPointLights) rtk_shadow_point(st, sc->dev, sc->hasCsg, count, coherent(level), coherent(level) ? sc->maxBlocksPacketShadow : sc->maxBlocks, ln->surfq, nsurf, sc->pointLights, sc->nPointLights, acc, aaSq, ln->ctr, false, sc->maxBlocksPacketShadow2);
				for (auto& al : sc->areaLights) rtk_shadow_area(st, sc->dev, sc->hasCsg, sc->nCsg, count, ln->surfq, nsurf, al.first, al.second, seed, acc, aaSq, ln->ctr);
				if (timeK) { HIPCHK(hipEventRecord(ln->ev[5], st)); shadowPending = true; }
				shadowRays += (uint64_t)nsurf * sc->shadowRaysPerSurface;
			}
			n = ln->ctrHost->next_count;
			cur ^= 1;
		}
		if (shadowPending) {
			float ms = 0;
			HIPCHK(hipEventSynchronize(ln->ev[5]));
			HIPCHK(hipEventElapsedTime(&ms, ln->ev[4], ln->ev[5])); stats.shadow_ms += ms;
		}
		// hits on reflective textured triangles that their own reflected ray's subtree came back to: lit with the last hit's texel
		rtk_texel_fix(st, sc->dev, sc->hasCsg, sc->texFix(*ln), texRecords, ln->ctr, acc, aaSq, seed, true);
		stats.rays_nearest += nearRays; stats.rays_shadow += shadowRays; stats.levels = std::max<uint64_t>(stats.levels, nLevels);
		return RT_OK;
	}

	// The same level loop running AHEAD of the host.  Above, every level costs a host round trip (the counts decide the next
	// launches), 35 - 80 us during which the lane's stream is empty: half the frame time of the bundled scenes, whose ray trees
	// are 17 levels of a few hundred thousand rays.  Here the kernels of level L read their item counts on the device (from
	// level L - 1's counter block) and are launched with bounds for their grids, so the host only has to know how far the ray
	// tree goes: it queues `ahead` levels beyond the last one whose counts it has read back, and stops when a level it reads
	// turns out empty (the levels already queued behind that one find zero items and leave at once).  Bounds: a hit has at
	// most one reflected and one transmitted child, so level L holds at most (children per hit)^k times the rays of level
	// L - k, and never more than the queue does.  Not for counting / per-kernel timing runs or the narration
	// (those read more than the counts after each level): they use levels().  Scenes that keep texel records do run ahead: the
	// records' counter is one per batch (RtTexFix::count) and k_publish reports it with every level.
	// Waits until k_publish has written the report queued with sequence number `seq`.  Spins for a while (the levels this loop is
	// for take tens of microseconds), then sleeps between looks; a GPU error or ten seconds without the report end the wait.
	// `level`: which level's report this is (for the message when it does not come).  Two ways to give up: the stream has gone idle
	// and ten seconds later the report is still missing (it was lost), or the stream is still busy after RT_LEVEL_TIMEOUT_S seconds
	// (default 120: a kernel of that level, or of one queued before it, is not finishing) -- the error then names the level and the
	// kernels that were queued for it, so that rt_last_error() says where the lane hung instead of the worker waiting for ever.
	int awaitReport(const RtLevelReport* slot, uint32_t seq, RtLevelReport& out, int level, bool packets) {
		static const uint32_t limitS = std::max<uint32_t>(1, envU32("RT_LEVEL_TIMEOUT_S", 120));
		const auto t0 = std::chrono::steady_clock::now();
		auto what = [&] {
			std::string k = packets ? "k_intersect_packet" : "k_intersect";
			k += ", k_shade";
			if (sc->hasTransmissive) k += ", k_transmit";
			k += ", k_publish";
			return "level " + std::to_string(level) + " of the ray tree (" + k + ")";
		};
		for (uint64_t spins = 0;; spins++) {
			if (__atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE) == seq) {
				out.next_count = slot->next_count; out.surf_count = slot->surf_count; out.overflow = slot->overflow; out.texrec_count = slot->texrec_count; out.state_cases = slot->state_cases; out.seq = seq;
				return RT_OK;
			}
			if (spins < 4000) { __builtin_ia32_pause(); continue; }
			if ((spins & 63) == 0) {
				const hipError_t e = hipStreamQuery(st);
				if (e != hipSuccess && e != hipErrorNotReady) { rt::set_error("waiting for the counts of " + what() + ": " + hipGetErrorString(e)); return RT_EHIP; }
				const auto waited = std::chrono::steady_clock::now() - t0;
				if (e == hipSuccess && __atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE) != seq && waited > std::chrono::seconds(10)) {
					rt::set_error("the counts of " + what() + " never arrived although the stream is idle");
					return RT_EHIP;
				}
				if (e == hipErrorNotReady && waited > std::chrono::seconds(limitS)) {
					rt::set_error("the stream is still busy " + std::to_string(limitS) + " s after queueing " + what() + ": a kernel of that level or of the shadow launches before it is not finishing (RT_LEVEL_TIMEOUT_S)");
					return RT_EHIP;
				}
			}
			std::this_thread::sleep_for(std::chrono::microseconds(spins < 8000 ? 5 : 50));
		}
	}
	bool canRunAhead() const {
		static const bool off = getenv("RT_NO_LEVELS_AHEAD") != nullptr;
		return !off && !(flags & (RT_FLAG_COUNTERS | RT_FLAG_TIME_KERNELS)) && !pick.ev;
	}
	int levelsAhead(uint32_t r0, uint32_t cnt) {
		static const int ahead = (int)std::max<uint32_t>(1, std::min<uint32_t>(8, envU32("RT_LEVELS_AHEAD", 2)));
		const uint32_t fan = (sc->hasReflective ? 1u : 0u) + (sc->hasTransmissive ? 1u : 0u);
		const int lastLevel = fan == 0 ? 0 : std::min(maxDepth, RT_MAX_LEVELS - 1);
		RtCounters* const L = ln->lvl;
		HIPCHK(hipMemsetAsync(L, 0, sizeof(RtCounters) * (size_t)(lastLevel + 1), st));
		if (src.camera) rtk_generate(st, src.cam, src.W, src.H, src.aaX, src.aaY, src.tileRank, src.tileRanks, r0, cnt, ln->ray[0]);
		else rtk_load_rays(st, src.a + 3 * (size_t)r0, src.b + 3 * (size_t)r0, cnt, 1, ln->ray[0], r0);
		const bool shadows = sc->shadowRaysPerSurface > 0;
		int queued = 0, read = 0;          // levels queued so far / levels whose counts the host has read
		uint64_t known = cnt;              // rays of level `read` (exact)
		uint64_t nearRays = 0, shadowRays = 0, nLevels = 0;
		uint32_t seqOf[RT_MAX_LEVELS] = {};
		texRecords = 0;
		bool overflow = false;
		int cur = 0;
		for (;;) {
			// queue levels while they are within `ahead` of what the host knows and can still hold rays
			while (queued <= lastLevel && queued < read + ahead && known > 0) {
				const int lv = queued;
				uint64_t bound = known;
				for (int k = read; k < lv; k++) bound = std::min<uint64_t>(bound * fan, sc->qcap);
				const uint32_t nb = (uint32_t)std::min<uint64_t>(bound, sc->qcap);
				const uint32_t* nDev = lv == 0 ? nullptr : &L[lv - 1].next_count;
				const bool packets = coherent(lv);
				// deeper levels (incoherent rays): one launch walks, shades and lights a ray (k_level); level 0 and scenes that keep texel records
				// go through the queues (RT_NO_FUSED_LEVELS: everything does -- read per call, the tests compare the two forms)
				// Level 0 of a SMALL scene's frame goes the same way (RT_NO_SMALL_LEVEL0: through the packet kernels): with the scene in LDS a
				// camera ray's walk is a few hundred instructions, while the persistent packet kernels, built for trees of 10^5 triangles, take
				// 60 us a launch whatever the scene (their launch tails; measured on the bundled scenes).
				const bool smallLevel0 = lv == 0 && src.camera && !src.outHit && sc->small.blob && aaSq <= envU32("RT_SMALL_LEVEL0_MAX_SPP", 4) && getenv("RT_NO_SMALL_LEVEL0") == nullptr && getenv("RT_NO_SMALL_SCENE") == nullptr;
				const bool fused = ((!packets && lv > 0) || smallLevel0) && !sc->texelFix && getenv("RT_NO_FUSED_LEVELS") == nullptr;
				// ... and the exit searches of transmitted rays as well where they are analytic (no CSG object, no triangle transmits): such a
				// level, like one of a scene without transmission, is a single launch that also reports its counts
				// Where they are not (CSG objects, meshes) they still go inline on levels small enough to be latency-bound -- a launch of a few
				// hundred waves lasts as long as its slowest lane either way, and a k_transmit behind it adds its own slowest exit search on top
				// (C3's deepest levels, scene l) -- while on large levels a few trapped rays would hold waves that have work waiting behind them.
				static const uint32_t inlineMax = envU32("RT_INLINE_TRANSMIT_MAX", 1048576);
				const bool inlineX = fused && sc->hasTransmissive && (sc->transmitAnalytic || nb <= inlineMax) && getenv("RT_NO_INLINE_TRANSMIT") == nullptr;
				const bool selfReport = fused && (!sc->hasTransmissive || inlineX) && getenv("RT_NO_SELF_REPORT") == nullptr;
				seqOf[lv] = ++ln->reportSeq ? ln->reportSeq : ++ln->reportSeq;   // (never 0)
				if (fused) rtk_level(st, sc->dev, sc->small, sc->hasCsg, inlineX, ln->ray[cur], nb, nDev, ln->ray[cur ^ 1], ln->surfq, ln->transq, &L[lv], acc, aaSq, maxDepth,
				                     !shadows ? 0 : sc->areaLights.empty() ? 1 : 2, sc->pointLights, (uint32_t)sc->nPointLights,
				                     sc->hasTransmissive ? &ln->ctr->state_cases : nullptr, selfReport ? ln->reportDev + lv : nullptr, seqOf[lv]);
				else {
				rtk_intersect(st, sc->dev, sc->hasCsg, false, packets, packets ? sc->maxBlocksPacketNear : sc->maxBlocks, ln->ray[cur], nb, ln->hitq, &L[lv], nDev);
				rtk_shade(st, sc->dev, sc->hasCsg, ln->ray[cur], ln->hitq, nb, ln->ray[cur ^ 1], ln->surfq, ln->transq, &L[lv], acc, aaSq, maxDepth,
				          shadows, lv == 0 ? src.outHit : nullptr, src.outId, src.outT, pick, sc->texFix(*ln), nDev);
				}
				if (sc->hasTransmissive && !inlineX) rtk_transmit(st, sc->dev, sc->hasCsg, ln->transq, nb, ln->ray[cur ^ 1], &L[lv], &ln->ctr->state_cases, acc, aaSq, maxDepth, pick);
				if (!selfReport) rtk_publish(st, &L[lv], sc->texelFix ? &ln->ctr->texrec_count : nullptr, sc->hasTransmissive ? &ln->ctr->state_cases : nullptr, ln->reportDev + lv, seqOf[lv]);
				if (shadows) {
					if (sc->nPointLights && !fused) rtk_shadow_point(st, sc->dev, sc->hasCsg, false, packets, packets ? sc->maxBlocksPacketShadow : sc->maxBlocks, ln->surfq, nb, sc->pointLights, sc->nPointLights, acc, aaSq, &L[lv], true, sc->maxBlocksPacketShadow2);
					for (auto& al : sc->areaLights) rtk_shadow_area(st, sc->dev, sc->hasCsg, sc->nCsg, false, ln->surfq, nb, al.first, al.second, seed, acc, aaSq, &L[lv], true);
				}
				if (int rcL = launchesOk()) return rcL;
				cur ^= 1;
				queued++;
			}
			if (read == queued) break;     // nothing in flight that the host has not seen
			// the counts of the oldest level not yet read
			RtLevelReport c;
			if (int rcW = awaitReport(ln->report + read, seqOf[read], c, read, coherent(read))) return rcW;
			nearRays += known;
			nLevels++;
			shadowRays += (uint64_t)std::min<uint32_t>(c.surf_count, sc->qcap) * sc->shadowRaysPerSurface;
			if (c.overflow) overflow = true;
			texRecords = c.texrec_count;   // (records of every level up to this one: k_publish runs behind the level's shading)
			stateCases = c.state_cases;    // (failed exit searches since begin(): k_publish runs behind the level's k_transmit)
			known = overflow ? 0 : std::min<uint32_t>(c.next_count, sc->qcap);
			read++;
			if (read > lastLevel) known = 0;
			if (known == 0) break;         // whatever is still queued behind this level finds no items and leaves at once
		}
		if (overflow) {
			// the levels still queued behind the one that overflowed may leave texel records too: batch() drops them by the count, so read it
			if (sc->texelFix) {
				HIPCHK(hipStreamSynchronize(st));
				HIPCHK(hipMemcpy(&texRecords, &ln->ctr->texrec_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
			}
			return 1;
		}
		// hits on reflective textured triangles that their own reflected ray's subtree came back to: lit with the last hit's texel
		rtk_texel_fix(st, sc->dev, sc->hasCsg, sc->texFix(*ln), texRecords, ln->ctr, acc, aaSq, seed, true);
		stats.rays_nearest += nearRays; stats.rays_shadow += shadowRays; stats.levels = std::max<uint64_t>(stats.levels, nLevels);
		return RT_OK;
	}

	// Runs [r0, r0+cnt); on a queue overflow discards what the batch accumulated and re-runs it in two halves.
	int batch(uint32_t r0, uint32_t cnt) {
		const rt_stats saved = stats;
		int rc = canRunAhead() ? levelsAhead(r0, cnt) : levels(r0, cnt);
		if (rc != 1) return rc;
		stats = saved;
		stats.batch_splits++;
		HIPCHK(hipMemsetAsync(&ln->ctr->overflow, 0, sizeof(uint32_t), st));
		rtk_texel_fix(st, sc->dev, sc->hasCsg, sc->texFix(*ln), texRecords, ln->ctr, acc, aaSq, seed, false);   // drop the discarded batch's records
		// clear this batch's pixels (resolve with no destination only zeroes the accumulators)
		rtk_resolve(st, acc, src.W, src.H, src.tileRank, src.tileRanks, r0 / aaSq, cnt / aaSq, aaSq, src.camera ? 0 : 1, nullptr);
		if (cnt <= aaSq) { rt::set_error("ray queue too small for the ray tree of a single pixel (raise RT_QUEUE_RAYS)"); return RT_ELIMIT; }
		const uint32_t half = std::max(aaSq, (cnt / 2 / aaSq) * aaSq);
		if (int r1 = batch(r0, half)) return r1;
		return batch(r0 + half, cnt - half);
	}

	int all(uint64_t totalRays) {
		const uint32_t perBatch = std::max<uint32_t>(aaSq, (sc->qcap / 2 / (256 * aaSq)) * 256 * aaSq);
		for (uint64_t r0 = 0; r0 < totalRays; r0 += perBatch) {
			const uint32_t cnt = (uint32_t)std::min<uint64_t>(perBatch, totalRays - r0);
			if (int rc = batch((uint32_t)r0, cnt)) return rc;
		}
		// (a batch that overflowed and was re-run in halves counted its failed searches twice: the count is then an upper bound)
		stats.state_cases = stateCases;
		return RT_OK;
	}
};

int begin(rt_scene* sc, rt_scene::Lane* ln, hipStream_t st, uint32_t flags) {
	HIPCHK(hipSetDevice(sc->device));
	HIPCHK(hipMemsetAsync(ln->ctr, 0, sizeof(RtCounters), st));
	HIPCHK(hipEventRecord(ln->ev[0], st));
	(void)flags;
	return RT_OK;
}
int finish(rt_scene* sc, rt_scene::Lane* ln, hipStream_t st, Run& run, rt_stats* stats) {
	if (int rcL = launchesOk()) return rcL;
	HIPCHK(hipEventRecord(ln->ev[1], st));
	if (run.flags & RT_FLAG_COUNTERS) HIPCHK(hipMemcpyAsync(ln->ctrHost, ln->ctr, sizeof(RtCounters), hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	HIPCHK(hipEventElapsedTime(&run.stats.gpu_ms, ln->ev[0], ln->ev[1]));
	if (run.flags & RT_FLAG_COUNTERS) {
		run.stats.nodes_nearest = ln->ctrHost->nodes_nearest; run.stats.prims_nearest = ln->ctrHost->prims_nearest;
		run.stats.nodes_shadow = ln->ctrHost->nodes_shadow; run.stats.prims_shadow = ln->ctrHost->prims_shadow;
		run.stats.steps_nearest = ln->ctrHost->steps_nearest; run.stats.steps_shadow = ln->ctrHost->steps_shadow;
		run.stats.max_walk = (float)ln->ctrHost->pad;   // longest nearest-hit walk of the call, in nodes (diagnostic)
	}
#if RT_PROF
	if (const char* path = getenv("RT_TRACE_FILE")) {   // raw dump of RtCounters::trace (tools/trace_summary.py)
		std::vector<RtCounters> c(1);
		HIPCHK(hipMemcpy(c.data(), ln->ctr, sizeof(RtCounters), hipMemcpyDeviceToHost));
		if (FILE* f = fopen(path, "wb")) { fwrite(c[0].trace, sizeof c[0].trace, 1, f); fwrite(c[0].waveLog, sizeof c[0].waveLog, 1, f); fclose(f); }
		for (int k = 0; k < 2; k++) { fprintf(stderr, "[lane histogram %s]", k ? "shadow" : "nearest"); for (int i = 0; i < 7; i++) fprintf(stderr, " %llu", c[0].laneHist[k][i]); fprintf(stderr, "\n"); }
	}
#endif
	if (stats) *stats = run.stats;
	return RT_OK;
}

int64_t localTiles(int W, int H, int rank, int ranks) {
	const int64_t T = (int64_t)((W + 31) / 32) * ((H + 7) / 8);
	return rank < T ? (T - rank + ranks - 1) / ranks : 0;
}
// pixels of the image that fall into this rank's tiles (edge tiles may be partial)
int64_t ownedPixels(int W, int H, int rank, int ranks) {
	const int64_t tilesX = (W + 31) / 32, T = tilesX * ((H + 7) / 8);
	int64_t n = 0;
	for (int64_t t = rank; t < T; t += ranks) {
		const int64_t tx = t % tilesX, ty = t / tilesX;
		n += std::min<int64_t>(32, W - tx * 32) * std::min<int64_t>(8, H - ty * 8);
	}
	return n;
}

// One shard of a frame on one lane (the lane's worker thread).
int renderShard(rt_scene* sc, rt_scene::Job* job, int k) {
	rt_scene::Lane* ln = &sc->lanes[k];
	// Lane 0 works on the caller's stream: the GPU exposes 4 hardware queues to a process, and a fifth stream in use
	// (the caller's, which at least carries the start and end events) made two lanes share one queue and run at half
	// speed.  Its previous shard may have gone to another stream: order them.
	hipStream_t st = ln->stream;
	if (k == 0 && getenv("RT_LANE0_OWN_STREAM") == nullptr) {
		st = job->st;
		if (sc->lane0Last && sc->lane0LastStream != st) HIPCHK(hipStreamWaitEvent(st, sc->lane0Last, 0));
		sc->lane0Last = job->evDone[0]; sc->lane0LastStream = st;
	}
	HIPCHK(hipStreamWaitEvent(st, job->evStart, 0));
	if (int rc = begin(sc, ln, st, job->flags)) return rc;
	// (a whole frame on this lane: the call's own tile split, the lane's own accumulators, results in slot 0)
	const int slot = job->whole ? 0 : k;
	const int vr = job->whole ? job->tileRank : job->tileRank + job->tileRanks * k, vn = job->whole ? job->tileRanks : job->tileRanks * job->nShards;
	const uint32_t aaSq = (uint32_t)(job->aaX * job->aaY);
	Run run{ sc, ln, st, {}, job->maxDepth, job->seed, job->flags, aaSq };
	run.use(job->whole ? sc->laneAccum[k] : sc->accum);
	run.src.camera = true; run.src.cam = job->cam; run.src.W = job->W; run.src.H = job->H; run.src.aaX = job->aaX; run.src.aaY = job->aaY;
	run.src.tileRank = vr; run.src.tileRanks = vn;
	const uint64_t localPix = (uint64_t)localTiles(job->W, job->H, vr, vn) * 256;
	if (int rc = run.all(localPix * aaSq)) return rc;
	rtk_resolve(st, run.acc, job->W, job->H, vr, vn, 0, (uint32_t)localPix, aaSq, 0, job->rgb);
	// the padding slots of partial edge tiles went through the first level as inactive lanes: they are not rays
	run.stats.rays_nearest -= (localPix - (uint64_t)ownedPixels(job->W, job->H, vr, vn)) * aaSq;
	if (int rcL = launchesOk()) return rcL;
	HIPCHK(hipEventRecord(job->evDone[slot], st));
	if (job->flags & (RT_FLAG_COUNTERS | RT_FLAG_TIME_KERNELS)) { if (int rc = finish(sc, ln, st, run, &job->part[slot])) return rc; }
	else job->part[slot] = run.stats;
	return RT_OK;
}

void laneWorker(rt_scene* sc, int k) {
	(void)hipSetDevice(sc->device);
	for (;;) {
		rt_scene::Job* job = nullptr;
		{
			std::unique_lock<std::mutex> g(sc->mtx);
			sc->cvWork.wait(g, [&] { return sc->stopping || !sc->laneQueue[k].empty(); });
			if (sc->laneQueue[k].empty()) return;   // stopping
			job = sc->laneQueue[k].front();
			sc->laneQueue[k].pop_front();
		}
		if (job->whole ? k == job->lane : k < job->nShards) {
			const int slot = job->whole ? 0 : k;
			static const bool dbg = getenv("RT_DEBUG_LANES") != nullptr;
			static const auto epoch = std::chrono::steady_clock::now();
			const double t0 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - epoch).count();
			job->rc[slot] = renderShard(sc, job, k);
			if (job->rc[slot]) job->err[slot] = rt_last_error();
			if (dbg) {
				(void)hipEventSynchronize(job->evDone[slot]);
				const double t1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - epoch).count();
				fprintf(stderr, "[lane %d] job %llu: %.0f .. %.0f us (%.0f)\n", k, (unsigned long long)job->id, t0, t1, t1 - t0);
			}
		}
		{
			std::lock_guard<std::mutex> g(sc->mtx);
			job->remaining--;
		}
		sc->cvDone.notify_all();
	}
}

// The most intervals a step of any CSG program of the scene can hold (stacked lists together, or the list an operator
// writes): a leaf gives at most 1, Union n1 + n2, Intersection n1 * n2 (csg.cpp:190-200 pairs every interval with every
// interval), Difference n1 * (n2 + 1) (csg.cpp:202-230: per interval of the first list one piece per overlapping
// interval of the second and one at the end; n1 + n2 if the lists are disjoint, which this does not rely on).  Kernels
// that keep the lists in LDS size them by this; clamped to the capacity of the general path, whose truncation then
// applies to both alike.
int csgCapacity(const rt_flat_scene* s) {
	long long need = 1;
	for (int32_t c = 0; c < s->n_csgs; c++) {
		std::vector<long long> stack;
		long long total = 0;
		for (int32_t k = 0; k < s->csgs[c].node_count; k++) {
			const int kind = s->csg_nodes[s->csgs[c].node_first + k].kind;
			if (kind >= RT_CSG_SPHERE) { stack.push_back(1); total += 1; }
			else if (stack.size() >= 2) {
				const long long n2 = stack.back(); stack.pop_back();
				const long long n1 = stack.back(); stack.pop_back();
				const long long n = std::min<long long>(kind == RT_CSG_INTERSECTION ? n1 * n2 : kind == RT_CSG_DIFFERENCE ? n1 * (n2 + 1) : n1 + n2, 1 << 20);
				need = std::max(need, n);
				total += n - n1 - n2;
				stack.push_back(n);
			}
			need = std::max(need, total);
		}
	}
	return (int)std::min<long long>(need, RTD_CSG_CAP);
}

struct DevBuf {   // scoped device allocation
	void* p = nullptr;
	~DevBuf() { if (p) (void)hipFree(p); }
	int alloc(size_t bytes) { HIPCHK(hipMalloc(&p, std::max<size_t>(bytes, 16))); return RT_OK; }
	template <class T> T* as() { return static_cast<T*>(p); }
};

}  // namespace

extern "C" {

// Test hook (tests/test_treelet_cut.py; no device needed): both treelet forms of a scene cut with `threads` threads (1 = the sequential
// cut), each reduced to a digest of the tree as a walk from the root sees it -- every record's boxes and aux words, every embedded
// triangle, every leaf reference, and where each reference to another record leads (its slot within the treelet and the record behind
// it) but not the treelets' numbers, which is all a parallel cut may change.  out[0..1] digests, out[2..3] treelets of the two forms.
int rt_debug_treelet_digest(const rt_flat_scene* s, int threads, uint64_t out[4]) {
	if (!s || !out || threads < 1) { rt::set_error("rt_debug_treelet_digest: bad argument"); return RT_EINVAL; }
	if (int rc = rt::validate_flat_scene(s)) return rc;
	AugTree aug;
	buildAugTree(s, true, aug);
	rt::PodVec<rt_bvh_node> binary, quads;
	bool runsB = false, runsQ = false;
	int quadDepth = 0;
	g_cutThreads = threads;
	const int lb = buildTreelets(s, aug, true, binary, runsB);
	const int lq = buildQuadTreelets(s, aug, true, quads, runsQ, quadDepth);
	g_cutThreads = 0;
	if (lb < 0 || lq < 0) return RT_EINVAL;
	auto mix = [](uint64_t& h, uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); h *= 0xff51afd7ed558ccdull; h ^= h >> 33; };
	auto mixBytes = [&](uint64_t& h, const void* p, size_t n) { const uint32_t* w = static_cast<const uint32_t*>(p); for (size_t i = 0; i < n / 4; i++) mix(h, w[i]); };
	bool bad = false;
	// binary form: a record = rt_bvh_node (two boxes, child[2], count[2])
	uint64_t hB = (uint64_t)lb << 1 | (runsB ? 1u : 0u);
	if (!binary.empty()) {
		std::vector<size_t> st{ 0 };
		while (!st.empty()) {
			const size_t at = st.back(); st.pop_back();
			if (at >= binary.size()) { bad = true; break; }
			const rt_bvh_node& n = binary[at];
			mixBytes(hB, n.box, sizeof n.box);
			for (int k = 1; k >= 0; k--) {
				const int32_t c = n.child[k];
				mix(hB, (uint32_t)n.count[k] & ~(RT_TREELET_SLOTS - 1u));   // (an embedded run's slot is hashed below, through its triangles)
				if (c >= 0) { const size_t i = (uint32_t)c & (RT_TREELET_CONS - 1u); mix(hB, (i % RT_TREELET_SLOTS) << 1 | (((uint32_t)c & RT_TREELET_CONS) ? 1u : 0u)); st.push_back(i); }
				else {
					mix(hB, (uint32_t)c);
					const uint32_t aux = (uint32_t)n.count[k];
					if (aux != RT_TREELET_NO_RUN && (aux >> 26) != 0 && (aux & RT_TREELET_EMBED)) {
						const size_t first = at / RT_TREELET_SLOTS * RT_TREELET_SLOTS + (aux & (RT_TREELET_SLOTS - 1u));
						if (first + (aux >> 26) > binary.size()) { bad = true; break; }
						mix(hB, aux & (RT_TREELET_SLOTS - 1u));
						mixBytes(hB, &binary[first], (aux >> 26) * sizeof(rt_bvh_node));
					}
				}
			}
			if (bad) break;
		}
	}
	// quad form: a record = 32 floats (four boxes, four children, four aux words) in two slots
	uint64_t hQ = (uint64_t)lq << 9 | (uint64_t)quadDepth << 1 | (runsQ ? 1u : 0u);
	if (!quads.empty() && !bad) {
		std::vector<size_t> st{ 0 };
		while (!st.empty()) {
			const size_t at = st.back(); st.pop_back();
			if (at + 1 >= quads.size()) { bad = true; break; }
			const float* rec = reinterpret_cast<const float*>(&quads[at]);
			const int32_t* child = reinterpret_cast<const int32_t*>(rec + 24);
			const uint32_t* aux = reinterpret_cast<const uint32_t*>(rec + 28);
			mixBytes(hQ, rec, 24 * sizeof(float));
			for (int k = 3; k >= 0; k--) {
				const int32_t c = child[k];
				if (c == RT_QUAD_EMPTY) { mix(hQ, 0x7fffffffu); continue; }
				if (c >= 0) { const size_t i = (uint32_t)c & (RT_TREELET_CONS - 1u); mix(hQ, (i % RT_QUAD_SLOTS) << 1 | (((uint32_t)c & RT_TREELET_CONS) ? 1u : 0u)); mix(hQ, aux[k]); st.push_back(i); }
				else {
					mix(hQ, (uint32_t)c); mix(hQ, aux[k]);
					if (aux[k] != RT_TREELET_NO_RUN && (aux[k] >> 26) != 0 && (aux[k] & RT_TREELET_EMBED)) {
						const size_t first = at / RT_QUAD_SLOTS * RT_QUAD_SLOTS + (aux[k] & (RT_QUAD_SLOTS - 1u));
						if (first + (aux[k] >> 26) > quads.size()) { bad = true; break; }
						mixBytes(hQ, &quads[first], (aux[k] >> 26) * sizeof(rt_bvh_node));
					}
				}
			}
			if (bad) break;
		}
	}
	if (bad) { rt::set_error("rt_debug_treelet_digest: a reference leads outside the treelets"); return RT_EINVAL; }
	out[0] = hB; out[1] = hQ; out[2] = binary.size() / RT_TREELET_SLOTS; out[3] = quads.size() / RT_QUAD_SLOTS;
	return RT_OK;
}

int rt_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int rt_scene_upload(const rt_flat_scene* s, int device, rt_scene** out) {
	if (!out) { rt::set_error("rt_scene_upload: out is NULL"); return RT_EINVAL; }
	*out = nullptr;
	if (int rc = rt::validate_flat_scene(s)) return rc;
	const int depth = treeDepth(s);
	if (depth > RT_MAX_TREE_DEPTH) { rt::set_error("BVH deeper than RT_MAX_TREE_DEPTH"); return RT_ELIMIT; }
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { rt::set_error("no HIP device available: the MI355X path has no CPU fallback"); return RT_EHIP; }
	if (device < 0 || device >= ndev) { rt::set_error("rt_scene_upload: no such device"); return RT_EINVAL; }
	HIPCHK(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, device));
	rt_scene* sc = new rt_scene();
	sc->device = device;
	if (int rcS = staticScratchFor(device, &sc->staticScratch)) { delete sc; return rcS; }
	sc->maxBlocks = std::max(1, prop.multiProcessorCount) * (int)std::max<uint32_t>(1, envU32("RT_BLOCKS_PER_CU", 8));
	sc->maxBlocksPacketNear = std::max(1, prop.multiProcessorCount) * (int)std::max<uint32_t>(1, envU32("RT_BLOCKS_PER_CU_NEAR", 2));
	sc->maxBlocksPacketShadow = std::max(1, prop.multiProcessorCount) * (int)std::max<uint32_t>(1, envU32("RT_BLOCKS_PER_CU_SHADOW", 3));
	sc->maxBlocksPacketShadow2 = std::max(1, prop.multiProcessorCount) * (int)std::max<uint32_t>(1, envU32("RT_BLOCKS_PER_CU_SHADOW2", 2));
	auto fail = [&](int rc) { rt_scene_free(sc); return rc; };
	DevScene& D = sc->dev;
	int rc;
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double tArr0 = now();
	if ((rc = sc->upload(s->planes, s->n_planes, &D.planes))) return fail(rc);
	if ((rc = sc->upload(s->spheres, s->n_spheres, &D.spheres))) return fail(rc);
	if ((rc = sc->upload(s->tris, s->n_tris, &D.tris))) return fail(rc);
	if ((rc = sc->upload(s->boxes, s->n_boxes, &D.boxes))) return fail(rc);
	if ((rc = sc->upload(s->csgs, s->n_csgs, &D.csgs))) return fail(rc);
	if ((rc = sc->upload(s->csg_nodes, s->n_csg_nodes, &D.csg_nodes))) return fail(rc);
	if ((rc = sc->upload(s->meshes, s->n_meshes, &D.meshes))) return fail(rc);
	if ((rc = sc->upload(s->mesh_tris, s->n_mesh_tris, &D.mesh_tris))) return fail(rc);
	if ((rc = sc->upload(s->uvs, s->n_uvs, &D.uvs))) return fail(rc);
	if ((rc = sc->upload(s->materials, s->n_materials, &D.materials))) return fail(rc);
	if ((rc = sc->upload(s->lights, s->n_lights, &D.lights))) return fail(rc);
	if ((rc = sc->upload(s->textures, s->n_textures, &D.textures))) return fail(rc);
	if ((rc = sc->upload(s->texels, s->n_texel_bytes, &D.texels))) return fail(rc);
	if ((rc = sc->upload(s->leaf_prims, s->n_leaf_prims, &D.leaf_prims))) return fail(rc);
	const bool timing = getenv("RT_LOAD_TIMING") != nullptr;
	size_t treeletBytes = 0; int treeDepth = 0;
	const double tUp0 = now();
	{
		MeshTrees mt;
		buildMeshTrees(s, mt);
		if ((rc = sc->upload(mt.nodes.data(), (int64_t)mt.nodes.size(), &D.mesh_nodes))) return fail(rc);
		if ((rc = sc->upload(mt.order.data(), (int64_t)mt.order.size(), &D.mesh_order))) return fail(rc);
		if ((rc = sc->upload(mt.roots.data(), (int64_t)mt.roots.size(), &D.mesh_root))) return fail(rc);
	}
	{
		static_assert(sizeof(rt_tri) == sizeof(rt_bvh_node) && sizeof(rt_bvh_node) == 64, "treelet slots are 64 B");
		rt::PodVec<rt_bvh_node> treelets, quads;
		bool externalRuns = false, quadRuns = false;
		int quadDepth = 0;
		AugTree aug;
		const double tA0 = now();
		buildAugTree(s, getenv("RT_NO_SPLIT") == nullptr, aug);
		const double tA1 = now();
		if (aug.depth > RT_STACK_DEPTH) buildAugTree(s, false, aug);   // (a leaf of > 4 * 2^20 triangles)
		// The two forms of the tree -- binary treelets for the per-lane walks, quad treelets for the packet walks -- are cut from the same
		// augmented tree independently of each other: the first on a thread of its own while this one cuts the second.
		const bool embed = getenv("RT_NO_EMBED") == nullptr;
		std::string treeletError;
		auto binary = std::async(std::launch::async, [&] {
			const int l = buildTreelets(s, aug, embed, treelets, externalRuns);
			if (l < 0) treeletError = rt_last_error();   // (the error text is per thread)
			return l;
		});
		int qlevels = buildQuadTreelets(s, aug, embed, quads, quadRuns, quadDepth);
		int levels = binary.get();
		if (levels < 0) { rt::set_error(treeletError); return fail(RT_EINVAL); }
		if (levels > RT_TREELET_MAX_LEVELS) {   // too deep a cut: both forms again over the tree without sub-trees, nothing embedded
			buildAugTree(s, false, aug);
			levels = buildTreelets(s, aug, false, treelets, externalRuns);
			if (levels >= 0) qlevels = buildQuadTreelets(s, aug, embed, quads, quadRuns, quadDepth);
		}
		if (qlevels > RT_TREELET_MAX_LEVELS) qlevels = buildQuadTreelets(s, aug, false, quads, quadRuns, quadDepth);
		if (levels < 0 || qlevels < 0) return fail(RT_EINVAL);
		if (levels > RT_TREELET_MAX_LEVELS || treelets.size() / RT_TREELET_SLOTS >= RT_TREELET_CONS / RT_TREELET_SLOTS) { rt::set_error("BVH too deep or too large for the treelet layout"); return fail(RT_ELIMIT); }
		if (qlevels > RT_TREELET_MAX_LEVELS || quads.size() / RT_QUAD_SLOTS >= RT_TREELET_CONS / RT_QUAD_SLOTS || !RT_PACKET_STACK_FITS(quadDepth)) {
			rt::set_error("BVH too deep or too large for the quad-node layout"); return fail(RT_ELIMIT);
		}
		const double tT0 = now();
		if (timing) fprintf(stderr, "[upload] arrays %.2f s, mesh trees %.2f s, sub-trees %.2f s, treelets and quads %.2f s: %zu treelets, %d levels, tree depth %d; %zu quad treelets, %d levels, depth %d\n",
		                    tUp0 - tArr0, tA0 - tUp0, tA1 - tA0, tT0 - tA1, treelets.size() / RT_TREELET_SLOTS, levels, aug.depth, quads.size() / RT_QUAD_SLOTS, qlevels, quadDepth);
		D.treelet_levels = levels;
		D.treelet_staging = (externalRuns || quadRuns) ? 1 : 0;
		D.quad_levels = qlevels;
		const rt_bvh_node* tl = nullptr;
		if ((rc = sc->upload(treelets.data(), (int64_t)treelets.size(), &tl))) return fail(rc);
		D.treelets = reinterpret_cast<const float4*>(tl);
		treeletBytes = treelets.size() * sizeof(rt_bvh_node); treeDepth = aug.depth;
		const rt_bvh_node* ql = nullptr;
		if ((rc = sc->upload(quads.data(), (int64_t)quads.size(), &ql))) return fail(rc);
		D.quads = reinterpret_cast<const float4*>(ql);
		if (timing) fprintf(stderr, "[upload] treelets and quads to the device %.2f s\n", now() - tT0);
		D.treelet_root = s->root_child < 0 ? s->root_child : 0;
		D.treelet_root_aux = (s->root_child < 0 && s->root_count > 0) ? externalAux(leafRun(s, s->root_child), s->root_child) : RT_TREELET_NO_RUN;
		if (D.treelet_root_aux != RT_TREELET_NO_RUN) D.treelet_staging = 1;
	}
	D.n_planes = s->n_planes; D.n_lights = s->n_lights;
	D.root_child = s->root_child; D.root_count = s->root_count;
	memcpy(D.root_box, s->root_box, sizeof D.root_box);
	memcpy(D.background, s->background, sizeof D.background);
	sc->hasCsg = s->n_csgs > 0;
	sc->nCsg = s->n_csgs;
	D.csg_cap = csgCapacity(s);
	D.csg_nodes_max = 1;
	for (int32_t i = 0; i < s->n_csgs; i++) D.csg_nodes_max = std::max(D.csg_nodes_max, s->csgs[i].node_count);
	for (int32_t i = 0; i < s->n_materials; i++) {
		const float* kt = s->materials[i].transmissive;
		if (kt[0] != 0 || kt[1] != 0 || kt[2] != 0) sc->hasTransmissive = true;
		const float* kr = s->materials[i].reflective;
		if (kr[0] != 0 || kr[1] != 0 || kr[2] != 0) sc->hasReflective = true;
	}

	sc->transmitAnalytic = sc->hasTransmissive && s->n_csgs == 0;   // (the CSG variants of k_level keep their exit searches in k_transmit whatever transmits)
	for (int32_t i = 0; i < s->n_csgs && sc->transmitAnalytic; i++) { const float* kt = s->materials[s->csgs[i].mat].transmissive; if (kt[0] != 0 || kt[1] != 0 || kt[2] != 0) sc->transmitAnalytic = false; }
	for (int64_t i = 0; i < s->n_tris && sc->transmitAnalytic; i++) { const float* kt = s->materials[s->tris[i].mat].transmissive; if (kt[0] != 0 || kt[1] != 0 || kt[2] != 0) sc->transmitAnalytic = false; }
	for (int64_t i = 0; i < s->n_tris && !sc->texelFix; i++) {
		const float* kr = s->materials[s->tris[i].mat].reflective;
		if (s->tris[i].uv >= 0 && (kr[0] != 0 || kr[1] != 0 || kr[2] != 0) && getenv("RT_NO_TEXEL_FIX") == nullptr) sc->texelFix = true;
	}

	std::vector<int32_t> pl;
	for (int i = 0; i < s->n_lights; i++) {
		const rt_light& L = s->lights[i];
		if (L.type == RT_LIGHT_DIRECTIONAL || L.type == RT_LIGHT_POINT || L.type == RT_LIGHT_SPOT) { pl.push_back(i); sc->shadowRaysPerSurface += 1; }
		else if ((L.type == RT_LIGHT_RECT || L.type == RT_LIGHT_CIRC) && L.samples > 0) { sc->areaLights.push_back({ i, (uint32_t)L.samples }); sc->shadowRaysPerSurface += (uint64_t)L.samples; }
	}
	sc->nPointLights = (uint32_t)pl.size();
	const int32_t* plDev = nullptr;
	if ((rc = sc->upload(pl.data(), (int64_t)pl.size(), &plDev))) return fail(rc);
	sc->pointLights = const_cast<int32_t*>(plDev);
	{	// small scenes (every scene the reference ships): what the per-lane walks and the shading read, as one blob for k_level's LDS copy
		struct Sec { int id; const void* dev; size_t bytes; };
		const Sec secs[] = {
			{ RT_SMALL_TREELETS, D.treelets, treeletBytes }, { RT_SMALL_LEAF_PRIMS, D.leaf_prims, (size_t)s->n_leaf_prims * sizeof(uint32_t) },
			{ RT_SMALL_SPHERES, D.spheres, (size_t)s->n_spheres * sizeof(rt_sphere) }, { RT_SMALL_BOXES, D.boxes, (size_t)s->n_boxes * sizeof(rt_box) },
			{ RT_SMALL_PLANES, D.planes, (size_t)s->n_planes * sizeof(rt_plane) }, { RT_SMALL_MATERIALS, D.materials, (size_t)s->n_materials * sizeof(rt_material) },
			{ RT_SMALL_LIGHTS, D.lights, (size_t)s->n_lights * sizeof(rt_light) }, { RT_SMALL_CSGS, D.csgs, (size_t)s->n_csgs * sizeof(rt_csg) },
			{ RT_SMALL_CSG_NODES, D.csg_nodes, (size_t)s->n_csg_nodes * sizeof(rt_csg_node) }, { RT_SMALL_LIGHT_LIST, plDev, pl.size() * sizeof(int32_t) },
			{ RT_SMALL_TRIS, D.tris, (size_t)s->n_tris * sizeof(rt_tri) } };
		const size_t stackRows = (size_t)std::max(2, treeDepth + 2), stackBytes = stackRows * RT_BLOCK * sizeof(int32_t);
		auto pad16 = [](size_t b) { return (b + 15) / 16 * 16; };
		size_t need = 0;
		for (const Sec& e : secs) if (e.id != RT_SMALL_TRIS) need += pad16(e.bytes);
		if (stackRows <= RT_STACK_DEPTH && stackBytes + need <= RT_SMALL_SCENE_LDS_BYTES && getenv("RT_NO_SMALL_SCENE_BLOB") == nullptr) {
			const bool withTris = stackBytes + need + pad16((size_t)s->n_tris * sizeof(rt_tri)) <= RT_SMALL_SCENE_LDS_BYTES;
			const size_t total = std::max<size_t>(16, need + (withTris ? pad16((size_t)s->n_tris * sizeof(rt_tri)) : 0));
			void* blob = nullptr;
			if (hipMalloc(&blob, total) != hipSuccess) { rt::set_error("hipMalloc failed (small-scene blob)"); return fail(RT_EHIP); }
			sc->allocs.push_back(blob);
			HIPCHK(hipMemset(blob, 0, total));
			size_t at = 0;
			for (const Sec& e : secs) {
				if (e.id == RT_SMALL_TRIS && !withTris) { sc->small.off[e.id] = 0xffffffffu; continue; }
				sc->small.off[e.id] = (uint32_t)at;
				if (e.bytes) HIPCHK(hipMemcpy((char*)blob + at, e.dev, e.bytes, hipMemcpyDeviceToDevice));
				at += pad16(e.bytes);
			}
			sc->small.blob = reinterpret_cast<const uint4*>(blob);
			sc->small.bytes = (uint32_t)total;
			sc->small.stackRows = (uint32_t)stackRows;
			if (timing) fprintf(stderr, "[upload] small scene: %zu B blob + %zu B stack for k_level's LDS\n", total, stackBytes);
		}
	}

	// Queue entries per lane: 8 Mi to begin with (1.5 GB a lane), grown on demand up to 32 Mi when a frame brings more rays per lane than
	// two batches hold (C5, 4K at 16 samples per pixel: 33 M primary rays a lane; 8 batches of 4 M leave every kernel's tail exposed eight
	// times: 41.4 ms a frame against 38.8 with 32 Mi entries, and 39.6 with 64 Mi -- the lanes then overlap less).  RT_QUEUE_RAYS fixes the size.
	sc->qcap = std::max<uint32_t>(1024, envU32("RT_QUEUE_RAYS", 8u << 20));
	sc->qcapMax = getenv("RT_QUEUE_RAYS") ? sc->qcap : std::max(sc->qcap, envU32("RT_QUEUE_RAYS_MAX", 32u << 20));
	sc->texCap = std::min<uint32_t>(sc->qcap, 2u << 20);
	const int nLanes = (int)std::min<uint32_t>(RT_MAX_LANES, std::max<uint32_t>(1, envU32("RT_LANES", 4)));   // the box exposes 4 hardware queues per process
	sc->lanes.resize(nLanes);
	// lane 0 runs on the caller's stream (renderShard); creating a stream it would not use would still claim one of
	// the process's 4 hardware queues and push a working lane onto a shared one
	for (int i = 0; i < nLanes; i++)
		if ((rc = sc->initLane(sc->lanes[i], i > 0 || getenv("RT_LANE0_OWN_STREAM") != nullptr))) return fail(rc);
	for (auto& j : sc->jobs) {
		bool ok = hipEventCreate(&j.evStart) == hipSuccess;
		for (int i = 0; i < nLanes && ok; i++) ok = hipEventCreate(&j.evDone[i]) == hipSuccess;
		if (!ok) { rt::set_error("hipEventCreate failed"); return fail(RT_EHIP); }
	}
	for (int i = 0; i < nLanes; i++) sc->workers.emplace_back([sc, i] { laneWorker(sc, i); });
	*out = sc;
	return RT_OK;
}

void rt_scene_free(rt_scene* sc) {
	if (!sc) return;
	(void)hipSetDevice(sc->device);
	{   // frames still in flight finish first: the lanes' workers use everything freed below
		std::unique_lock<std::mutex> g(sc->mtx);
		sc->drainLocked(g);
		sc->stopping = true;
	}
	sc->cvWork.notify_all();
	for (auto& t : sc->workers) t.join();
	(void)hipDeviceSynchronize();
	for (void* p : sc->allocs) (void)hipFree(p);
	if (sc->accum.rgb) (void)hipFree(sc->accum.rgb);
	if (sc->accum.nonfinite) (void)hipFree(sc->accum.nonfinite);
	for (RtAccum& a : sc->laneAccum) { if (a.rgb) (void)hipFree(a.rgb); if (a.nonfinite) (void)hipFree(a.nonfinite); }
	if (sc->texHeads) (void)hipFree(sc->texHeads);
	for (auto& ln : sc->lanes) {
		if (ln.ctrHost) (void)hipHostFree(ln.ctrHost);
		if (ln.report) (void)hipHostFree(ln.report);
		for (auto& ev : ln.ev) if (ev) (void)hipEventDestroy(ev);
		if (ln.stream) (void)hipStreamDestroy(ln.stream);
	}
	for (auto& j : sc->jobs) {
		if (j.evStart) (void)hipEventDestroy(j.evStart);
		for (auto& e : j.evDone) if (e) (void)hipEventDestroy(e);
	}
	delete sc;
}

int rt_render_submit(rt_scene* sc, const rt_camera* camera, int width, int height, int aa_x, int aa_y, int max_depth, int tile_rank, int tile_ranks,
                     uint32_t seed, uint32_t flags, float* rgb_dev, void* stream, uint64_t* ticket) {
	if (!sc || !camera || !rgb_dev || !ticket || width <= 0 || height <= 0 || aa_x < 1 || aa_x > 16 || aa_y < 1 || aa_y > 16 || max_depth < 0 || max_depth > 30 ||
	    tile_ranks < 1 || tile_rank < 0 || tile_rank >= tile_ranks || (int64_t)width * height * (aa_x * aa_y) >= (1ll << 31)) {
		rt::set_error("rt_render_submit: bad argument");
		return RT_EINVAL;
	}
	*ticket = 0;
	HIPCHK(hipSetDevice(sc->device));
	hipStream_t st = static_cast<hipStream_t>(stream);
	// shards: one per lane for a frame-sized job; one when per-kernel timings or counters are asked for (they are only
	// meaningful for kernels running alone) or when the job is small
	const bool exclusive = (flags & (RT_FLAG_COUNTERS | RT_FLAG_TIME_KERNELS)) != 0;
	const int64_t tiles = localTiles(width, height, tile_rank, tile_ranks);
	int nShards = (int)sc->lanes.size();
	if (exclusive || tiles < 64 * nShards) nShards = 1;
	// RT_FLAG_WHOLE_FRAME: the frame on one lane (not for scenes that keep texel records: their per-pixel lists are shared like the accumulators)
	const bool whole = (flags & RT_FLAG_WHOLE_FRAME) != 0 && !exclusive && !sc->texelFix && sc->lanes.size() > 1;
	if (whole) nShards = 1;

	std::unique_lock<std::mutex> g(sc->mtx);
	// any free slot (tickets may be waited for in any order); the ticket carries its slot: ticket % RT_MAX_INFLIGHT
	rt_scene::Job* job = nullptr;
	for (rt_scene::Job& j : sc->jobs) if (j.id == 0) { job = &j; break; }
	if (!job) { rt::set_error("rt_render_submit: too many frames in flight (wait for an earlier ticket first)"); return RT_EINVAL; }
	const uint64_t newTicket = sc->nextId * RT_MAX_INFLIGHT + (uint64_t)(job - sc->jobs);
	// Shards of consecutive frames may overlap only if they own the same pixels of the shared accumulators
	// Whole frames go to the lanes with streams of their own, 1 .. n-1, in turn.  Lane 0 works on the caller's stream, and rt_render_wait makes that
	// stream wait for every frame it retires (work the caller queues afterwards must see the frame): a whole frame on lane 0 would sit behind
	// the other lanes' frames.  With the caller's stream carrying only those events the process stays within its four hardware queues.
	const int lane = whole ? 1 + sc->nextWholeLane++ % ((int)sc->lanes.size() - 1) : 0;
	// (a whole frame has its lane's accumulators to itself -- the lane's stream orders it behind that lane's earlier work -- so it overlaps
	// with anything unless those accumulators must grow; sharded frames overlap with each other as before and with whole frames freely)
	bool compatible = !exclusive && (whole ? sc->laneAccum[lane].npix >= (uint32_t)(width * height) && sc->laneAccum[lane].rgb != nullptr
	                                       : sc->accum.npix >= (uint32_t)(width * height) && sc->accum.rgb);
	for (const rt_scene::Job& j : sc->jobs)
		if (j.id && j.remaining > 0 && ((j.flags & (RT_FLAG_COUNTERS | RT_FLAG_TIME_KERNELS)) ||
		    (!whole && !j.whole && (j.W != width || j.H != height || j.tileRank != tile_rank || j.tileRanks != tile_ranks || j.nShards != nShards))))
			compatible = false;
	// queues for this frame's share per lane: a batch is half a queue (its rays' children must fit beside them)
	{
		const uint64_t share = ((uint64_t)tiles * 256u /* 32 x 8 pixels */ * (uint64_t)(aa_x * aa_y) + (uint64_t)nShards - 1) / (uint64_t)nShards;
		uint32_t want = sc->qcap;
		while (want < sc->qcapMax && (uint64_t)want < 2 * share) want = (uint32_t)std::min<uint64_t>(2ull * want, sc->qcapMax);
		if (want > sc->qcap) {
			sc->drainLocked(g);
			if (int rc = sc->growQueues(want)) return rc;
		}
	}
	if (!compatible) sc->drainLocked(g);
	if (int rc = whole ? sc->ensureLaneAccum(lane, (uint32_t)(width * height)) : sc->ensureAccum((uint32_t)(width * height))) return rc;
	job->cam = *camera; job->W = width; job->H = height; job->aaX = aa_x; job->aaY = aa_y; job->maxDepth = max_depth;
	job->tileRank = tile_rank; job->tileRanks = tile_ranks; job->nShards = nShards; job->seed = seed; job->flags = flags; job->rgb = rgb_dev; job->st = st;
	job->whole = whole; job->lane = lane;
	HIPCHK(hipEventRecord(job->evStart, st));   // the lanes start after whatever the caller queued on its stream before this call
	job->id = newTicket;
	sc->nextId++;
	job->remaining = (int)sc->lanes.size();
	for (size_t k = 0; k < sc->lanes.size(); k++) { job->rc[k] = RT_OK; job->err[k].clear(); job->part[k] = rt_stats{}; sc->laneQueue[k].push_back(job); }
	*ticket = job->id;
	g.unlock();
	sc->cvWork.notify_all();
	return RT_OK;
}

int rt_render_wait(rt_scene* sc, uint64_t ticket, rt_stats* stats) {
	if (!sc || ticket == 0) { rt::set_error("rt_render_wait: bad argument"); return RT_EINVAL; }
	HIPCHK(hipSetDevice(sc->device));
	rt_scene::Job* job = &sc->jobs[ticket % RT_MAX_INFLIGHT];
	{
		std::unique_lock<std::mutex> g(sc->mtx);
		if (job->id != ticket) { rt::set_error("rt_render_wait: unknown ticket"); return RT_EINVAL; }
		sc->cvDone.wait(g, [&] { return job->remaining == 0; });
	}
	auto retire = [&] { std::lock_guard<std::mutex> g(sc->mtx); job->id = 0; };
	for (int k = 0; k < job->nShards; k++)
		if (job->rc[k]) {
			const int rc = job->rc[k];
			rt::set_error(job->err[k]);
			for (auto& ln : sc->lanes) (void)hipStreamSynchronize(ln.stream);
			retire();
			return rc;
		}
	// The caller's stream is made to wait for the shards (work queued on it afterwards sees the frame); the host waits for
	// the shards' own events only -- the stream may already hold later frames' work (lane 0 runs on it) and the caller's
	// collectives, which this frame does not depend on.
	hipError_t e = hipSuccess;
	rt_stats total{};
	for (int k = 0; k < job->nShards && e == hipSuccess; k++) e = hipStreamWaitEvent(job->st, job->evDone[k], 0);
	for (int k = 0; k < job->nShards && e == hipSuccess; k++) {
		e = hipEventSynchronize(job->evDone[k]);
		float ms = 0;
		if (e == hipSuccess) e = hipEventElapsedTime(&ms, job->evStart, job->evDone[k]);
		total.gpu_ms = std::max(total.gpu_ms, ms);
	}
	for (int k = 0; k < job->nShards; k++) {
		const rt_stats& p = job->part[k];
		total.rays_nearest += p.rays_nearest; total.rays_shadow += p.rays_shadow; total.batch_splits += p.batch_splits;
		total.nodes_nearest += p.nodes_nearest; total.prims_nearest += p.prims_nearest; total.nodes_shadow += p.nodes_shadow; total.prims_shadow += p.prims_shadow;
		total.nearest_ms += p.nearest_ms; total.shadow_ms += p.shadow_ms; total.steps_nearest += p.steps_nearest; total.steps_shadow += p.steps_shadow;
		total.levels = std::max(total.levels, p.levels); total.max_walk = std::max(total.max_walk, p.max_walk); total.state_cases += p.state_cases;
	}
	retire();
	if (e != hipSuccess) { rt::set_error(std::string("rt_render_wait: ") + hipGetErrorString(e)); return RT_EHIP; }
	if (stats) *stats = total;
	return RT_OK;
}

int rt_render(rt_scene* sc, const rt_camera* camera, int width, int height, int aa_x, int aa_y, int max_depth, int tile_rank, int tile_ranks,
              uint32_t seed, uint32_t flags, float* rgb_dev, void* stream, rt_stats* stats) {
	uint64_t ticket = 0;
	if (int rc = rt_render_submit(sc, camera, width, height, aa_x, aa_y, max_depth, tile_rank, tile_ranks, seed, flags, rgb_dev, stream, &ticket)) return rc;
	return rt_render_wait(sc, ticket, stats);
}

int rt_render_host(rt_scene* sc, const rt_camera* camera, int width, int height, int aa_x, int aa_y, int max_depth, uint32_t seed, uint32_t flags,
                   float* rgb_host, rt_stats* stats) {
	if (!sc || !rgb_host || width <= 0 || height <= 0) { rt::set_error("rt_render_host: bad argument"); return RT_EINVAL; }
	HIPCHK(hipSetDevice(sc->device));
	DevBuf frame;
	const size_t bytes = (size_t)width * height * 3 * sizeof(float);
	if (int rc = frame.alloc(bytes)) return rc;
	if (int rc = rt_render(sc, camera, width, height, aa_x, aa_y, max_depth, 0, 1, seed, flags, frame.as<float>(), nullptr, stats)) return rc;
	HIPCHK(hipMemcpy(rgb_host, frame.p, bytes, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_trace_batch(rt_scene* sc, const float* e, const float* s, size_t n, int max_depth, uint32_t seed, uint32_t flags,
                   uint8_t* hit, int32_t* obj_id, float* t, float* rgb, rt_stats* stats) {
	if (!sc || (n > 0 && (!e || !s)) || max_depth < 0 || max_depth > 30 || n > (1ull << 31)) { rt::set_error("rt_trace_batch: bad argument"); return RT_EINVAL; }
	if (stats) memset(stats, 0, sizeof *stats);
	if (n == 0) return RT_OK;
	sc->drain();   // the lanes' queues are used directly below
	HIPCHK(hipSetDevice(sc->device));
	DevBuf de, ds, dhit, did, dt, drgb;
	if (int rc = de.alloc(n * 12)) return rc;
	if (int rc = ds.alloc(n * 12)) return rc;
	if (int rc = dhit.alloc(n)) return rc;
	if (int rc = did.alloc(n * 4)) return rc;
	if (int rc = dt.alloc(n * 4)) return rc;
	if (int rc = drgb.alloc(n * 12)) return rc;
	HIPCHK(hipMemcpy(de.p, e, n * 12, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(ds.p, s, n * 12, hipMemcpyHostToDevice));
	hipStream_t st = nullptr;
	rt_scene::Lane* ln = &sc->lanes[0];
	if (int rc = begin(sc, ln, st, flags)) return rc;
	if (int rc = sc->ensureAccum((uint32_t)n)) return rc;
	Run run{ sc, ln, st, {}, max_depth, seed, flags, 1u };
	run.use(sc->accum);
	run.src.a = de.as<float>(); run.src.b = ds.as<float>();
	run.src.outHit = dhit.as<uint8_t>(); run.src.outId = did.as<int32_t>(); run.src.outT = dt.as<float>();
	if (int rc = run.all(n)) return rc;
	rtk_resolve(st, sc->accum, 0, 0, 0, 1, 0, (uint32_t)n, 1, 1, drgb.as<float>());
	if (int rc = finish(sc, ln, st, run, stats)) return rc;
	if (hit) HIPCHK(hipMemcpy(hit, dhit.p, n, hipMemcpyDeviceToHost));
	if (obj_id) HIPCHK(hipMemcpy(obj_id, did.p, n * 4, hipMemcpyDeviceToHost));
	if (t) HIPCHK(hipMemcpy(t, dt.p, n * 4, hipMemcpyDeviceToHost));
	if (rgb) HIPCHK(hipMemcpy(rgb, drgb.p, n * 12, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_trace_pick(rt_scene* sc, const float* e, const float* s, int max_depth, uint32_t seed, uint8_t* hit, float* rgb, rt_pick_event* events, size_t cap, size_t* n_events) {
	if (!sc || !e || !s || max_depth < 0 || max_depth > 30 || (cap > 0 && !events) || cap > (1u << 24)) { rt::set_error("rt_trace_pick: bad argument"); return RT_EINVAL; }
	if (n_events) *n_events = 0;
	sc->drain();   // lane 0's queues are used directly below
	HIPCHK(hipSetDevice(sc->device));
	DevBuf de, ds, dhit, did, dt, drgb, dev, dcount;
	if (int rc = de.alloc(12)) return rc;
	if (int rc = ds.alloc(12)) return rc;
	if (int rc = dhit.alloc(1)) return rc;
	if (int rc = did.alloc(4)) return rc;
	if (int rc = dt.alloc(4)) return rc;
	if (int rc = drgb.alloc(12)) return rc;
	if (int rc = dev.alloc(std::max<size_t>(cap, 1) * sizeof(rt_pick_event))) return rc;
	if (int rc = dcount.alloc(4)) return rc;
	HIPCHK(hipMemcpy(de.p, e, 12, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(ds.p, s, 12, hipMemcpyHostToDevice));
	HIPCHK(hipMemset(dcount.p, 0, 4));
	hipStream_t st = nullptr;
	rt_scene::Lane* ln = &sc->lanes[0];
	if (int rc = begin(sc, ln, st, 0)) return rc;
	if (int rc = sc->ensureAccum(1)) return rc;
	Run run{ sc, ln, st, {}, max_depth, seed, 0u, 1u };
	run.use(sc->accum);
	run.src.a = de.as<float>(); run.src.b = ds.as<float>();
	run.src.outHit = dhit.as<uint8_t>(); run.src.outId = did.as<int32_t>(); run.src.outT = dt.as<float>();
	run.pick = RtPick{ dev.as<rt_pick_event>(), dcount.as<uint32_t>(), (uint32_t)cap };
	if (int rc = run.all(1)) return rc;
	rtk_resolve(st, sc->accum, 0, 0, 0, 1, 0, 1u, 1, 1, drgb.as<float>());
	if (int rc = finish(sc, ln, st, run, nullptr)) return rc;
	uint32_t count = 0;
	HIPCHK(hipMemcpy(&count, dcount.p, 4, hipMemcpyDeviceToHost));
	if (n_events) *n_events = count;
	if (events && count > 0) HIPCHK(hipMemcpy(events, dev.p, std::min<size_t>(count, cap) * sizeof(rt_pick_event), hipMemcpyDeviceToHost));
	if (hit) HIPCHK(hipMemcpy(hit, dhit.p, 1, hipMemcpyDeviceToHost));
	if (rgb) HIPCHK(hipMemcpy(rgb, drgb.p, 12, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_intersect_batch(rt_scene* sc, const float* e, const float* d, size_t n, uint32_t flags, uint8_t* hit, int32_t* obj_id, float* t, rt_stats* stats) {
	if (!sc || (n > 0 && (!e || !d)) || n > (1ull << 31)) { rt::set_error("rt_intersect_batch: bad argument"); return RT_EINVAL; }
	if (stats) memset(stats, 0, sizeof *stats);
	if (n == 0) return RT_OK;
	sc->drain();   // the lanes' queues are used directly below
	HIPCHK(hipSetDevice(sc->device));
	DevBuf de, dd, dhit, did, dt;
	if (int rc = de.alloc(n * 12)) return rc;
	if (int rc = dd.alloc(n * 12)) return rc;
	if (int rc = dhit.alloc(n)) return rc;
	if (int rc = did.alloc(n * 4)) return rc;
	if (int rc = dt.alloc(n * 4)) return rc;
	HIPCHK(hipMemcpy(de.p, e, n * 12, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(dd.p, d, n * 12, hipMemcpyHostToDevice));
	hipStream_t st = nullptr;
	rt_scene::Lane* ln = &sc->lanes[0];
	if (int rc = begin(sc, ln, st, flags)) return rc;
	Run run{ sc, ln, st, {}, 0, 0, flags, 1u };
	run.use(sc->accum);
	for (size_t c0 = 0; c0 < n; c0 += sc->qcap) {
		const uint32_t cnt = (uint32_t)std::min<size_t>(sc->qcap, n - c0);
		HIPCHK(hipMemsetAsync(ln->ctr, 0, RT_COUNTER_RESET_BYTES, st));
		rtk_load_rays(st, de.as<float>() + 3 * c0, dd.as<float>() + 3 * c0, cnt, 0, ln->ray[0], 0);
		rtk_intersect(st, sc->dev, sc->hasCsg, (flags & RT_FLAG_COUNTERS) != 0, false, sc->maxBlocks, ln->ray[0], cnt, ln->hitq, ln->ctr);
		rtk_export_hits(st, sc->dev, ln->hitq, cnt, dhit.as<uint8_t>() + c0, did.as<int32_t>() + c0, dt.as<float>() + c0);
		run.stats.rays_nearest += cnt;
	}
	run.stats.levels = 1;
	if (int rc = finish(sc, ln, st, run, stats)) return rc;
	if (hit) HIPCHK(hipMemcpy(hit, dhit.p, n, hipMemcpyDeviceToHost));
	if (obj_id) HIPCHK(hipMemcpy(obj_id, did.p, n * 4, hipMemcpyDeviceToHost));
	if (t) HIPCHK(hipMemcpy(t, dt.p, n * 4, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_intersect_batch_static(rt_scene* sc, const float* e_dev, const float* d_dev, size_t n, uint8_t* hit_dev, int32_t* obj_id_dev, float* t_dev, void* stream) {
	if (!sc || (n > 0 && (!e_dev || !d_dev))) { rt::set_error("rt_intersect_batch_static: bad argument"); return RT_EINVAL; }
	if (n > RT_STATIC_MAX_RAYS) { rt::set_error("rt_intersect_batch_static: more than RT_STATIC_MAX_RAYS rays in one call (the scratch is a fixed device array; split the batch)"); return RT_ELIMIT; }
	if (n == 0) return RT_OK;
	HIPCHK(hipSetDevice(sc->device));
	hipStream_t st = static_cast<hipStream_t>(stream);
	// Under stream capture nothing here may wait for, or ask about, device work: the drain below blocks the host on the events of frames
	// in flight, and the first call on a device loads the kernels and asks the runtime for their occupancy.  So a capturing caller must
	// have no frame in flight and must have made one ordinary call on this handle before (both refused with RT_EINVAL otherwise); what
	// is left -- one memset and three launches on `stream` -- is what the graph records.
	hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
	if (st) HIPCHK(hipStreamIsCapturing(st, &capture));
	if (capture == hipStreamCaptureStatusActive) {
		std::unique_lock<std::mutex> g(sc->mtx);
		bool busy = false;
		for (const rt_scene::Job& j : sc->jobs) busy |= j.id != 0;
		if (busy) { rt::set_error("rt_intersect_batch_static: frames are in flight (or not yet waited for) while the stream is capturing; wait for them first"); return RT_EINVAL; }
		if (!sc->staticWarm) { rt::set_error("rt_intersect_batch_static: the first call on a scene handle loads its kernels and cannot be captured; make one call outside the capture first"); return RT_EINVAL; }
	} else
	sc->drain();   // frames in flight use the same traversal kernels' cursors only through their own lanes, but the call promises "after outstanding frames"
	StaticScratch* ws = sc->staticScratch;   // device address (hipGetSymbolAddress), never the array's name
	const RtQueue rq{ ws->ray, RT_STATIC_MAX_RAYS }, hq{ ws->hit, RT_STATIC_MAX_RAYS };
	// the array keeps the last call's contents: the work cursor and the queue counts must start at zero; the queue slots
	// [0, n) are all rewritten by k_load_rays / k_intersect before they are read
	HIPCHK(hipMemsetAsync(&ws->ctr, 0, sizeof(RtCounters), st));
	rtk_load_rays(st, e_dev, d_dev, (uint32_t)n, 0, rq, 0);
	rtk_intersect(st, sc->dev, sc->hasCsg, false, false, sc->maxBlocks, rq, (uint32_t)n, hq, &ws->ctr);
	rtk_export_hits(st, sc->dev, hq, (uint32_t)n, hit_dev, obj_id_dev, t_dev);
	if (int rcL = launchesOk()) return rcL;
	if (capture != hipStreamCaptureStatusActive) { std::lock_guard<std::mutex> g(sc->mtx); sc->staticWarm = true; }
	return RT_OK;
}

int rt_shadow_batch(rt_scene* sc, const float* point, const float* light_pos, size_t n, uint32_t flags, uint8_t* lit, float* shadow, rt_stats* stats) {
	if (!sc || (n > 0 && (!point || !light_pos)) || n > (1ull << 31)) { rt::set_error("rt_shadow_batch: bad argument"); return RT_EINVAL; }
	if (stats) memset(stats, 0, sizeof *stats);
	if (n == 0) return RT_OK;
	sc->drain();   // the lanes' queues are used directly below
	HIPCHK(hipSetDevice(sc->device));
	DevBuf dp, dl, dlit, dsh;
	if (int rc = dp.alloc(n * 12)) return rc;
	if (int rc = dl.alloc(n * 12)) return rc;
	if (int rc = dlit.alloc(n)) return rc;
	if (int rc = dsh.alloc(n * 12)) return rc;
	HIPCHK(hipMemcpy(dp.p, point, n * 12, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(dl.p, light_pos, n * 12, hipMemcpyHostToDevice));
	hipStream_t st = nullptr;
	rt_scene::Lane* ln = &sc->lanes[0];
	if (int rc = begin(sc, ln, st, flags)) return rc;
	Run run{ sc, ln, st, {}, 0, 0, flags & ~RT_FLAG_COUNTERS, 1u };
	run.use(sc->accum);
	rtk_shadow_query(st, sc->dev, sc->hasCsg, dp.as<float>(), dl.as<float>(), (uint32_t)n, dlit.as<uint8_t>(), dsh.as<float>());
	run.stats.rays_shadow = n;
	if (int rc = finish(sc, ln, st, run, stats)) return rc;
	if (lit) HIPCHK(hipMemcpy(lit, dlit.p, n, hipMemcpyDeviceToHost));
	if (shadow) HIPCHK(hipMemcpy(shadow, dsh.p, n * 12, hipMemcpyDeviceToHost));
	return RT_OK;
}

int64_t rt_tiles_count(int width, int height, int tile_rank, int tile_ranks) {
	if (width <= 0 || height <= 0 || tile_ranks < 1 || tile_rank < 0 || tile_rank >= tile_ranks) return -1;
	return localTiles(width, height, tile_rank, tile_ranks);
}

static int tilesCopy(rt_scene* sc, const float* src, float* dst, int width, int height, int tile_rank, int tile_ranks, void* stream, int mode) {
	if (!sc || !src || !dst || rt_tiles_count(width, height, tile_rank, tile_ranks) < 0) { rt::set_error("rt_tiles_*: bad argument"); return RT_EINVAL; }
	HIPCHK(hipSetDevice(sc->device));
	hipStream_t st = static_cast<hipStream_t>(stream);
	// enqueued, not waited for: between a rendered frame and its gather the host must stay free to submit the next frame
	rtk_tiles_copy(st, src, dst, width, height, tile_rank, tile_ranks, (uint32_t)(localTiles(width, height, tile_rank, tile_ranks) * 256), mode);
	if (int rcL = launchesOk()) return rcL;
	return RT_OK;
}
int rt_tiles_pack(rt_scene* sc, const float* frame_dev, float* packed_dev, int width, int height, int tile_rank, int tile_ranks, void* stream) {
	return tilesCopy(sc, frame_dev, packed_dev, width, height, tile_rank, tile_ranks, stream, 0);
}
int rt_tiles_unpack(rt_scene* sc, const float* packed_dev, float* frame_dev, int width, int height, int tile_rank, int tile_ranks, void* stream) {
	return tilesCopy(sc, packed_dev, frame_dev, width, height, tile_rank, tile_ranks, stream, 1);
}
int rt_tiles_merge(rt_scene* sc, const float* src_dev, float* dst_dev, int width, int height, int tile_rank, int tile_ranks, void* stream) {
	return tilesCopy(sc, src_dev, dst_dev, width, height, tile_rank, tile_ranks, stream, 2);
}

static int tilesRgb8(rt_scene* sc, const void* src, uint8_t* dst, int width, int height, int tile_rank, int tile_ranks, void* stream, int mode) {
	if (!sc || !src || !dst || rt_tiles_count(width, height, tile_rank, tile_ranks) < 0) { rt::set_error("rt_tiles_*_rgb8: bad argument"); return RT_EINVAL; }
	HIPCHK(hipSetDevice(sc->device));
	rtk_tiles_rgb8(static_cast<hipStream_t>(stream), src, dst, width, height, tile_rank, tile_ranks, (uint32_t)(localTiles(width, height, tile_rank, tile_ranks) * 256), mode);
	if (int rcL = launchesOk()) return rcL;
	return RT_OK;
}
int rt_tiles_pack_rgb8(rt_scene* sc, const float* frame_dev, uint8_t* packed_dev, int width, int height, int tile_rank, int tile_ranks, void* stream) {
	return tilesRgb8(sc, frame_dev, packed_dev, width, height, tile_rank, tile_ranks, stream, 0);
}
int rt_tiles_unpack_rgb8(rt_scene* sc, const uint8_t* packed_dev, uint8_t* frame8_dev, int width, int height, int tile_rank, int tile_ranks, void* stream) {
	return tilesRgb8(sc, packed_dev, frame8_dev, width, height, tile_rank, tile_ranks, stream, 1);
}
int rt_tiles_quantize_rgb8(rt_scene* sc, const float* frame_dev, uint8_t* frame8_dev, int width, int height, int tile_rank, int tile_ranks, void* stream) {
	return tilesRgb8(sc, frame_dev, frame8_dev, width, height, tile_rank, tile_ranks, stream, 2);
}

}  // extern "C"
