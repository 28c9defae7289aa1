// The wavefront engine's persistent late-bounce kernel on the CPU: bounce_tail_block (pt_wavefront.hpp, what k_bounce_tail
// runs) executed as a block of 4 waves x 64 host threads through tests/host/hip_shim.  The queue entries are RAYS of bounce
// `entry` (a rays file as for extend_wave_check, each with a throughput, a stream and a path id of its own); every path's
// end -- lit flag, radiance, (sequential streams) stream word -- and the rays per bounce are compared with the same paths
// run ONE LANE at a time by the plain loop bvh_intersect + shade_lane, i.e. with what the staged extend / shade stages do.
//   bounce_tail_check <blob> <pt_frame file> <rays file: N x {ox oy oz dx dy dz best_t} float32> <entry bounce> [lanes=64]
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "pt_wavefront.hpp"

using namespace ptd;

static std::vector<unsigned char> read_file(const char* path) {
	std::vector<unsigned char> v;
	FILE* f = std::fopen(path, "rb");
	if (!f) { std::perror(path); std::exit(2); }
	std::fseek(f, 0, SEEK_END);
	long n = std::ftell(f);
	std::fseek(f, 0, SEEK_SET);
	v.resize((size_t)n);
	if (n && std::fread(v.data(), 1, (size_t)n, f) != (size_t)n) { std::perror("read"); std::exit(2); }
	std::fclose(f);
	return v;
}
template <typename T> static T* exact(size_t n) {
	T* p = (T*)std::malloc((n ? n : 1) * sizeof(T));
	std::memset(p, 0xDB, (n ? n : 1) * sizeof(T));
	return p;
}

int main(int argc, char** argv) {
	if (argc < 5) { std::fprintf(stderr, "usage: bounce_tail_check blob frame rays entry_bounce [lanes]\n"); return 2; }
	std::vector<unsigned char> blob_v = read_file(argv[1]), frame_v = read_file(argv[2]), rays_v = read_file(argv[3]);
	if (frame_v.size() != sizeof(pt_frame)) { std::fprintf(stderr, "frame file is %zu bytes, pt_frame is %zu\n", frame_v.size(), sizeof(pt_frame)); return 2; }
	pt_frame fr;
	std::memcpy(&fr, frame_v.data(), sizeof(fr));
	const uint32_t entry = (uint32_t)std::atol(argv[4]);
	const int lanes = argc > 5 ? std::atoi(argv[5]) : 64;
	unsigned char* blob = (unsigned char*)std::malloc(blob_v.size());
	std::memcpy(blob, blob_v.data(), blob_v.size());
	const SceneHeader* hdr = reinterpret_cast<const SceneHeader*>(blob);
	if (hdr->magic != kSceneMagic || hdr->blob_bytes != blob_v.size() || hdr->n_nodes == 0) { std::fprintf(stderr, "not a BVH scene blob\n"); return 2; }
	const FrameDev f = to_dev(fr);
	if (entry == 0 || entry >= f.depth) { std::fprintf(stderr, "entry bounce must be 1 .. depth-1\n"); return 2; }
	const bool sequential = fr.rng_mode == PT_RNG_SEQUENTIAL;
	const uint32_t n_tris = hdr->n_tris, n_mats = hdr->n_mats;
	const float4* tris = reinterpret_cast<const float4*>(blob + sizeof(SceneHeader));
	const float4* mats = tris + (size_t)n_tris * 4;
	const BvhView bv = make_bvh_view(blob, n_tris, n_mats, f.origin);
	const float* rays = reinterpret_cast<const float*>(rays_v.data());
	const uint32_t count = (uint32_t)(rays_v.size() / (7 * sizeof(float)));
	// one block (G = 1): entry j sits in slot j; path ids are a permutation of the slots
	float4 *qa = exact<float4>(count), *qb = exact<float4>(count), *qc = exact<float4>(count);
	uint32_t* qpid = exact<uint32_t>(count);
	for (uint32_t k = 0; k < count; k++) {
		const float* r = rays + 7 * k;
		uint32_t rng = (k * 2654435761u) | 1u;
		qa[k] = float4{ r[0], r[1], r[2], r[3] };
		qb[k] = float4{ r[4], r[5], r[6], __uint_as_float(rng) };
		qc[k] = float4{ 0.25f + (k % 7) * 0.1f, 0.5f, 1.0f - (k % 5) * 0.15f, 1.0f };
		qpid[k] = (uint32_t)(((unsigned long long)k * 7919u + 13u) % count);         // 7919 is prime: a permutation unless it divides count
	}
	if (count % 7919u == 0) { std::fprintf(stderr, "ray count must not be a multiple of 7919\n"); return 2; }
	static const ByteJump jump31 = make_byte_jump(31);
	uint32_t* s_jump = exact<uint32_t>(1024);
	for (uint32_t k = 0; k < 1024u; k++) s_jump[k] = (&jump31.t[0][0])[k];
	const bool det_bounded = (hdr->flags & kSceneDetBounded) != 0;

	// ---- the reference: one lane, the plain loop
	float4* ref_rad = exact<float4>(count);
	unsigned char* ref_lit = exact<unsigned char>(count);
	uint32_t* ref_rng = exact<uint32_t>(count);
	std::vector<unsigned long long> ref_rays(f.depth + 2, 0);
	{
		std::vector<uint32_t> stack((size_t)bvh_stack_depth(n_tris) * 256, 0xDEADBEEFu);
		ShadeCtx c{ f, tris, mats, s_jump, n_tris, det_bounded, count, ref_rad, ref_lit, sequential ? ref_rng : nullptr, nullptr, nullptr };
		const bool extended = f.shading == PT_SHADING_EXTENDED;
		for (uint32_t k = 0; k < count; k++) {
			float4 la = qa[k], lb = qb[k], lc = qc[k];
			uint32_t pid = qpid[k];
			for (uint32_t bounce = entry;; bounce++) {
				float best = lb.z;
				uint32_t tri = kNoHit;
				bvh_intersect(bv, Vec3{ la.x, la.y, la.z }, Vec3{ la.w, lb.x, lb.y }, best, tri, stack.data());
				PathNext nx;
				nx.pid = pid;
				const uint32_t last = bounce + 1u >= f.depth ? 1u : 0u;
				shade_lane<false, false, false, false>(c, extended, last, 0u, uint2{ __float_as_uint(best), tri }, float4{ 0, 0, 0, 0 }, &la, &lb, &lc, nx);
				if (nx.cont) ref_rays[bounce + 1]++;
				if (!nx.cont || last) break;
				la = float4{ nx.o.x, nx.o.y, nx.o.z, nx.d.x };
				lb = float4{ nx.d.y, nx.d.z, nx.t, __uint_as_float(nx.rng) };
				lc = nx.step;
			}
		}
	}

	// ---- the block
	float4* rad = exact<float4>(count);
	unsigned char* lit = exact<unsigned char>(count);
	uint32_t* rng_out = exact<uint32_t>(count);
	uint32_t* s_stack = exact<uint32_t>((size_t)bvh_stack_depth(n_tris) * 256);
	uint32_t s_next = 0, s_count[PT_MAX_DEPTH + 1] = {};
	WaveShim waves[4];
	for (auto& w : waves) w.lanes = lanes;
	std::vector<std::thread> th;
	for (unsigned w = 0; w < 4; w++)
		for (int l = 0; l < lanes; l++)
			th.emplace_back([&, w, l] {
				threadIdx.x = w * 64 + (unsigned)l;
				t_wave = &waves[w];
				ShadeCtx c{ f, tris, mats, s_jump, n_tris, det_bounded, count, rad, lit, sequential ? rng_out : nullptr, nullptr, nullptr };
				bounce_tail_block(c, bv, entry, qa, qb, qc, qpid, count, s_stack, &s_next, s_count);
			});
	for (auto& t : th) t.join();

	size_t bad = 0, lit_paths = 0;
	for (uint32_t p = 0; p < count; p++) {
		bool ok = lit[p] == ref_lit[p] && (!sequential || rng_out[p] == ref_rng[p]);
		if (ok && lit[p] == 1) ok = std::memcmp(&rad[p], &ref_rad[p], sizeof(float4)) == 0;
		lit_paths += ref_lit[p] == 1;
		if (!ok) { if (bad < 10) std::fprintf(stderr, "path %u: lit %u (reference %u)\n", p, lit[p], ref_lit[p]); bad++; }
	}
	unsigned long long later = 0;
	for (uint32_t b = entry + 1; b <= f.depth; b++) {
		later += ref_rays[b];
		if (s_count[b] != ref_rays[b]) { std::fprintf(stderr, "bounce %u: %u rays counted, reference %llu\n", b, s_count[b], ref_rays[b]); bad++; }
	}
	std::printf("paths %u lit %zu later_rays %llu mismatches %zu lanes %d\n", count, lit_paths, later, bad, lanes);
	return bad ? 1 : 0;
}
